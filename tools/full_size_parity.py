#!/usr/bin/env python3
"""BASELINE configs[1] at FULL size (1920x1080x256 spp, full BSDF set, 530 841 600 samples, 506.25 generations) on the
GPU and on the CPU oracle (watertight mode): integer event totals and image RMS.  ~5 min of CPU on a 16-core box, so this
is a tool (its output is kept under profiles/), not a test.

usage: full_size_parity.py [spp] [scene]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from oracle.oracle import Oracle  # noqa: E402
from rtcuda_amd import api, scenes  # noqa: E402

w, h = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variant = sys.argv[2] if len(sys.argv) > 2 else "full_bsdf"
arrays = scenes.cornell_bunny(variant)
gpu = api.Scene(arrays)
out = {"frame": f"{variant} {w}x{h}x{spp}", "samples": w * h * spp}
pairs = (("shade_events", "sum_mat"), ("any_rays", "sum_ah"), ("emission_adds", "emission_adds"),
         ("shadow_adds", "ah_adds"), ("rr_draws", "rr_draws"))
imgs = {}
for name, env in (("k_paths", {}),):
    for k, v in env.items():
        os.environ[k] = v
    img, st = gpu.render(api.make_camera(aspect=w / h), w, h, spp)
    for k in env:
        del os.environ[k]
    imgs[name] = img
    out[name] = {g: int(st[g]) for g, _ in pairs}
    out[name]["seconds_render"] = st["seconds_render"]
    print(name, out[name], flush=True)
orc = Oracle("pinned")
osc = orc.scene(arrays).set_watertight(True)
t = time.time()
cores = min(os.cpu_count() or 8, 16)
oimg, _, ost = osc.render(orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h), w, h, spp, threads=cores)
out["oracle_watertight"] = {g: int(ost[o]) for g, o in pairs}
out["oracle_seconds"] = time.time() - t
for name in ("k_paths",):
    a = imgs[name]
    m = ~(np.isnan(a) | np.isnan(oimg))
    out[name]["events_equal"] = all(out[name][g] == out["oracle_watertight"][g] for g, _ in pairs)
    out[name]["rms"] = float(np.sqrt(np.mean((a[m].astype(np.float64) - oimg[m]) ** 2)))
    out[name]["max_abs"] = float(np.abs(a[m] - oimg[m]).max())
    out[name]["nan_pixels"] = [int(np.isnan(a).any(axis=2).sum()), int(np.isnan(oimg).any(axis=2).sum())]
print(json.dumps(out, indent=1))
