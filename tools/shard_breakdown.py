#!/usr/bin/env python3
"""Where the time of one slot shard goes: wall, render loop, the persistent kernel, lockstep rounds.

usage: shard_breakdown.py [shard_count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rtcuda_amd import api, scenes
w, h, spp = 1920, 1080, 256
scene = api.Scene(scenes.cornell_bunny("full_bsdf"))
cam = api.make_camera(aspect=w / h)
fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene.render_shard(cam, w, h, spp, 0, R, fb.data_ptr())
for rep in range(2):
    fb.zero_(); torch.cuda.synchronize()
    t = time.perf_counter()
    st = scene.render_shard(cam, w, h, spp, 0, R, fb.data_ptr(), flags=api.FLAG_TIME_KERNELS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"R={R} wall {dt*1e3:.2f} ms render {st['seconds_render']*1e3:.2f} ms k_paths {st['seconds_trace']*1e3:.2f} ms iterations {st['iterations']} rate {st['camera_rays']/dt/1e6:.1f}")
