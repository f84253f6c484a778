#!/usr/bin/env python3
"""Frame time of the opt-in parity mode RT_FLAG_REFERENCE_WALK (never the timed kernels) on two full BASELINE frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rtcuda_amd import api, scenes
w, h, spp = 1920, 1080, 256
for variant in ("full_bsdf", "sixteen_lights"):
    sc = api.Scene(scenes.cornell_bunny(variant))
    cam = api.make_camera(aspect=w / h)
    fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    for env in ({},):
        for k, v in env.items(): os.environ[k] = v
        for rep in range(2):
            fb.zero_(); torch.cuda.synchronize(); t = time.perf_counter()
            st = sc.render_shard(cam, w, h, spp, 0, 1, fb.data_ptr(), flags=api.FLAG_REFERENCE_WALK)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        for k in env: del os.environ[k]
        print(variant, env, f"{dt*1e3:.1f} ms", st["shade_events"], st["shadow_adds"], flush=True)
