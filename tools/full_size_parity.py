#!/usr/bin/env python3
"""BASELINE frames at FULL size (1920x1080, 256 / 512 / 1024 spp) on the GPU and on the CPU oracle, float images side by side:

  literal      the reference's own tree, fp32 slab test and tree-order tie rule (bvh.cuh:30-357, aabb_intersector.cuh:14-36,
               triangle.cuh:49) against the DEFAULT kernels (round 5: they make the reference's decisions): the strict
               comparison -- integer event totals equal, image RMS at the float-atomics noise, no pixel over 1e-4
  watertight   conservative box decisions + the caller-order tie rule against RT_FLAG_WATERTIGHT (also strict), and that
               frame against the LITERAL oracle: what the reference's lost hits amount to on a whole frame

Minutes of CPU per frame and mode on a 16-core box, so this is a tool (its output is kept under profiles/), not a test; the
GPU suite holds the same frames to committed event totals and fixed-point image hashes.

usage: full_size_parity.py [spp] [scene] [--no-watertight]"""
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
do_watertight = "--no-watertight" not in sys.argv
arrays = scenes.cornell_bunny(variant)
gpu = api.Scene(arrays)
out = {"frame": f"{variant} {w}x{h}x{spp}", "samples": w * h * spp, "build_id": api.build_id()}
pairs = (("shade_events", "sum_mat"), ("any_rays", "sum_ah"), ("emission_adds", "emission_adds"),
         ("shadow_adds", "ah_adds"), ("rr_draws", "rr_draws"))
img, st = gpu.render(api.make_camera(aspect=w / h), w, h, spp)
out["k_paths_default"] = {g: int(st[g]) for g, _ in pairs}
out["k_paths_default"].update(seconds_render=st["seconds_render"], literal_retraces=st["literal_retraces"],
                              reference_lost_hits=st["reference_lost_hits"], exact_ties=st["exact_ties"])
print("default kernels", out["k_paths_default"], flush=True, file=sys.stderr)
orc = Oracle("pinned")
cores = usable_cpus()
cam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)


def compare(a, b):
    m = ~(np.isnan(a) | np.isnan(b))
    d = np.abs(a.astype(np.float64) - b)
    return {"rms": float(np.sqrt(np.mean(d[m] ** 2))), "max_abs": float(d[m].max()),
            "pixels_over_1e-4": int((np.nan_to_num(d).max(axis=2) > 1e-4).sum()),
            "nan_pixels_gpu_oracle": [int(np.isnan(a).any(axis=2).sum()), int(np.isnan(b).any(axis=2).sum())],
            "nan_pixels_same": bool(np.array_equal(np.isnan(a), np.isnan(b)))}


osc = orc.scene(arrays)
t = time.time()
limg, _, lst = osc.render(cam, w, h, spp, threads=cores)
out["oracle_literal"] = {g: int(lst[o]) for g, o in pairs}
out["oracle_literal"]["seconds"] = time.time() - t
out["default_vs_literal"] = compare(img, limg)
out["default_vs_literal"]["events_equal"] = all(out["k_paths_default"][g] == out["oracle_literal"][g] for g, _ in pairs)
out["default_vs_literal"]["within_north_star_1e-4_rms"] = out["default_vs_literal"]["rms"] < 1e-4
print("default vs literal", out["default_vs_literal"], flush=True, file=sys.stderr)
if do_watertight:
    wimg_g, wst_g = gpu.render(api.make_camera(aspect=w / h), w, h, spp, flags=api.FLAG_WATERTIGHT)
    out["k_paths_watertight_flag"] = {g: int(wst_g[g]) for g, _ in pairs}
    t = time.time()
    oimg, _, ost = osc.set_watertight(True).render(cam, w, h, spp, threads=cores)
    out["oracle_watertight"] = {g: int(ost[o]) for g, o in pairs}
    out["oracle_watertight"]["seconds"] = time.time() - t
    out["watertight_flag_vs_watertight"] = compare(wimg_g, oimg)
    out["watertight_flag_vs_watertight"]["events_equal"] = all(out["k_paths_watertight_flag"][g] == out["oracle_watertight"][g] for g, _ in pairs)
    out["watertight_flag_vs_literal"] = compare(wimg_g, limg)
    out["watertight_flag_vs_literal"]["event_deltas"] = {g: out["k_paths_watertight_flag"][g] - out["oracle_literal"][g] for g, _ in pairs}
    print("watertight flag vs watertight", out["watertight_flag_vs_watertight"], flush=True, file=sys.stderr)
    print("watertight flag vs literal", out["watertight_flag_vs_literal"], flush=True, file=sys.stderr)
out["cores"] = cores
print(json.dumps(out, indent=1))
