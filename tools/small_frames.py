#!/usr/bin/env python3
"""Frame time of small frames (a single generation: nothing but the lockstep rounds) and of the full C2 frame, best of 5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rtcuda_amd import api, scenes
sc = api.Scene(scenes.cornell_bunny("full_bsdf"))
for (w, h, spp) in ((256, 256, 4), (640, 360, 4), (1920, 1080, 256)):
    cam = api.make_camera(aspect=w / h)
    fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    ts = []
    for r in range(6):
        fb.zero_(); torch.cuda.synchronize(); t = time.perf_counter()
        st = sc.render_shard(cam, w, h, spp, 0, 1, fb.data_ptr())
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(os.environ.get("RT_LIB_NAME", "librtcuda_amd.so"), (w, h, spp), "best %.3f ms" % (1e3 * min(ts[1:])), "iterations", st["iterations"],
          "k_paths %.3f" % (1e3 * st["seconds_trace"]), "render %.3f" % (1e3 * st["seconds_render"]), st["camera_rays"], st["shade_events"], flush=True)
