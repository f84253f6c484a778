#!/usr/bin/env python3
"""BASELINE frames at FULL size (1920x1080, 256 / 512 / 1024 spp) on the GPU and on the CPU oracle, in BOTH oracle modes:

  watertight   conservative box decisions + the caller-order tie rule: what the product implements -> the strict
               comparison (integer event totals equal, image RMS)
  literal      the reference's own fp32 slab test and tree-order tie rule (aabb_intersector.cuh:14-36, triangle.cuh:49):
               event deltas, RMS and pixels over 1e-4 -- the reference's walk loses about one accepted hit in 10^7 rays
               (tests/test_traversal_audit.py), and this records what that amounts to on a whole frame

Minutes of CPU per frame and mode on a 16-core box, so this is a tool (its output is kept under profiles/), not a test.

usage: full_size_parity.py [spp] [scene] [--no-literal]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from oracle.oracle import Oracle, usable_cpus  # noqa: E402
from rtcuda_amd import api, scenes  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
w, h = 1920, 1080
spp = int(args[0]) if len(args) > 0 else 256
variant = args[1] if len(args) > 1 else "full_bsdf"
do_literal = "--no-literal" not in sys.argv
arrays = scenes.cornell_bunny(variant)
gpu = api.Scene(arrays)
out = {"frame": f"{variant} {w}x{h}x{spp}", "samples": w * h * spp, "build_id": api.build_id()}
pairs = (("shade_events", "sum_mat"), ("any_rays", "sum_ah"), ("emission_adds", "emission_adds"),
         ("shadow_adds", "ah_adds"), ("rr_draws", "rr_draws"))
img, st = gpu.render(api.make_camera(aspect=w / h), w, h, spp)
out["k_paths"] = {g: int(st[g]) for g, _ in pairs}
out["k_paths"]["seconds_render"] = st["seconds_render"]
print("k_paths", out["k_paths"], flush=True, file=sys.stderr)
orc = Oracle("pinned")
cores = usable_cpus()
cam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)


def compare(oimg):
    m = ~(np.isnan(img) | np.isnan(oimg))
    d = np.abs(img.astype(np.float64) - oimg)
    return {"rms": float(np.sqrt(np.mean(d[m] ** 2))), "max_abs": float(d[m].max()),
            "pixels_over_1e-4": int((np.nan_to_num(d).max(axis=2) > 1e-4).sum()),
            "nan_pixels_gpu_oracle": [int(np.isnan(img).any(axis=2).sum()), int(np.isnan(oimg).any(axis=2).sum())],
            "nan_pixels_same": bool(np.array_equal(np.isnan(img), np.isnan(oimg)))}


osc = orc.scene(arrays).set_watertight(True)
t = time.time()
oimg, _, ost = osc.render(cam, w, h, spp, threads=cores)
out["oracle_watertight"] = {g: int(ost[o]) for g, o in pairs}
out["oracle_watertight"]["seconds"] = time.time() - t
out["vs_watertight"] = compare(oimg)
out["vs_watertight"]["events_equal"] = all(out["k_paths"][g] == out["oracle_watertight"][g] for g, _ in pairs)
print("watertight", out["vs_watertight"], flush=True, file=sys.stderr)
if do_literal:
    t = time.time()
    limg, _, lst = osc.set_watertight(False).render(cam, w, h, spp, threads=cores)
    out["oracle_literal"] = {g: int(lst[o]) for g, o in pairs}
    out["oracle_literal"]["seconds"] = time.time() - t
    out["vs_literal_reference_walk"] = compare(limg)
    out["vs_literal_reference_walk"]["event_deltas_gpu_minus_literal"] = {
        g: out["k_paths"][g] - out["oracle_literal"][g] for g, _ in pairs}
    out["vs_literal_reference_walk"]["within_north_star_1e-4_rms"] = out["vs_literal_reference_walk"]["rms"] < 1e-4
    print("literal", out["vs_literal_reference_walk"], flush=True, file=sys.stderr)
out["cores"] = cores
print(json.dumps(out, indent=1))
