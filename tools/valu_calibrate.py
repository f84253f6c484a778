#!/usr/bin/env python3
"""Vector-ALU calibration: what a pure, independent v_fma_f32 stream sustains at 1, 2 and 4 waves per SIMD.

Run plainly it prints the rates; run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` its k_valu_calibrate dispatches show what those counters read at a KNOWN issue rate
-- which settles how `VALUBusy = 4 x SQ_ACTIVE_INST_VALU / (SIMDs x cycles)` of the render kernels is to be read.

Short bursts on purpose (1.6 ms per launch, best of six): a pure FMA stream held for many milliseconds runs into the
power limit and the clock drops (100 000 iterations: 44 - 52 instead of 55 T lane-op/s at 4 waves per SIMD); the render
kernel itself runs at 2.4 GHz for its whole 160 ms (GRBM_GUI_ACTIVE / duration)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (brings the HIP runtime of the process)
from rtcuda_amd import api  # noqa: E402

out = {}
for waves in (1, 2, 4, 8):
    rate, winstr = api.calibrate_valu(waves, 20000)
    out[f"waves_per_simd_{waves}"] = {"lane_ops_per_s": rate, "T_lane_ops_per_s": round(rate / 1e12, 2),
                                      "frac_of_spec_peak": round(rate / (256 * 4 * 32 * 2.4e9), 4),
                                      "wave_instructions_per_launch": winstr}
# Packed fp32 (round 4): the same 16 chains on register pairs.  `x_scalar` = lane-operations/s relative to the v_fma_f32
# stream at the same occupancy: 2.0 means a packed instruction issues at the rate of a scalar one.
for kind, name in ((1, "v_pk_fma_f32"), (2, "v_pk_mul_f32"), (3, "v_pk_add_f32")):
    for waves in (1, 2, 4, 8):
        rate = api.calibrate_valu_packed(waves, 20000, kind)
        base = out[f"waves_per_simd_{waves}"]["lane_ops_per_s"]
        out[f"{name}_waves_per_simd_{waves}"] = {"T_lane_ops_per_s": round(rate / 1e12, 2), "x_scalar": round(rate / base, 3)}
print(json.dumps(out))
