#!/usr/bin/env python3
"""What ONE wave can issue: cycles per instruction of a wave, by instruction kind and by the number of waves on its SIMD.

k_paths' blocks take the same time whether the SIMD holds one wave or four, so the kernel is bound by the issue rate of a
single wave; this table says what that rate is for the instructions the kernel is made of (rt_probe_issue: 16 independent
chains per lane, inline asm).  cycles = launch seconds x 2.4e9 / instructions per wave."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (brings the HIP runtime of the process)
from rtcuda_amd import api  # noqa: E402

CLOCK = 2.4e9
out = {}
for kind, name in enumerate(api.PROBE_ISSUE_KINDS):
    row = {}
    for waves in (1, 2, 4, 8):
        sec, n = api.probe_issue(kind, waves, 20000)
        row[f"w{waves}"] = round(sec * CLOCK / n, 2)  # cycles per instruction of ONE wave at this occupancy
    row["simd_cycles_per_instr_w4"] = round(row["w4"] / 4, 2)
    row["simd_cycles_per_instr_w8"] = round(row["w8"] / 8, 2)
    out[name] = row
    print(f"{name:42s} 1 wave {row['w1']:6.2f}  2 waves {row['w2']:6.2f}  4 waves {row['w4']:6.2f}  8 waves {row['w8']:6.2f}   "
          f"(SIMD: {row['simd_cycles_per_instr_w4']:.2f} / {row['simd_cycles_per_instr_w8']:.2f} cycles per instruction at 4 / 8 waves)", flush=True)
print(json.dumps(out))
