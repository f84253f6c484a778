#!/usr/bin/env python3
"""How much of a traversal block's duration is the size of the scene?  Renders the C2 frame with every k-th bunny
triangle only (k = 1, 2, 4, 8, 16: the BVH and the triangle records shrink by k, the tree loses log2(k) levels, the
rays stay about as long) -- with the instrumented library (make -C rtcuda_amd/csrc prof; RT_LIB_NAME=librtcuda_amd_prof.so)
the per-block cycle counts of k_paths appear on stderr.  A measurement tool, not a parity case (the thinned bunny has holes).

usage: RT_LIB_NAME=librtcuda_amd_prof.so python tools/working_set_probe.py [--spp 64]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rtcuda_amd import api, scenes  # noqa: E402

spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 64
w, h = 1920, 1080
full = scenes.cornell_bunny("full_bsdf")
n_bunny = full.n_tris - 12  # the driver's recipe: bunny faces first, then 10 wall and 2 light triangles
cam = api.make_camera(aspect=w / h)
fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
for k in (1, 2, 4, 8, 16):
    keep = np.concatenate([np.arange(0, n_bunny, k), np.arange(n_bunny, full.n_tris)])
    remap = -np.ones(full.n_tris, dtype=np.int64)
    remap[keep] = np.arange(len(keep))
    lights = full.lights.copy()
    lights["tri"] = np.where(lights["tri"] >= 0, remap[np.maximum(lights["tri"], 0)], -1).astype(np.int32)
    arrays = scenes.SceneArrays(tris=full.tris[keep].copy(), tri_material=full.tri_material[keep].copy(), tri_light=full.tri_light[keep].copy(),
                                materials=full.materials, lights=lights, name=f"full_bsdf/{k}", meta=full.meta)
    sc = api.Scene(arrays)
    print(f"---- every {k}-th bunny triangle: {len(keep)} triangles", file=sys.stderr, flush=True)
    for r in range(2):
        fb.zero_()
        st = sc.render_shard(cam, w, h, spp, 0, 1, fb.data_ptr(), flags=api.FLAG_TIME_KERNELS)
        torch.cuda.synchronize()
    print(f"every {k}-th: {len(keep)} triangles, kernel {st['seconds_trace'] * 1e3:.2f} ms, any {st['any_rays']}, shades {st['shade_events']}", flush=True)
    sc.close()
