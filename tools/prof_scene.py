#!/usr/bin/env python3
"""One frame of a BASELINE scene, timed; with RT_LIB_NAME=librtcuda_amd_prof.so (make -C rtcuda_amd/csrc prof) the
instrumented kernel prints where the waves' cycles go (block counts, lanes per block) on stderr.

usage: prof_scene.py [scene] [spp] [shard_count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rtcuda_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "full_bsdf"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w, h = 1920, 1080
scene = api.Scene(scenes.cornell_bunny(name))
cam = api.make_camera(aspect=w / h)
fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
for rep in range(2):
    fb.zero_(); torch.cuda.synchronize()
    t = time.perf_counter()
    st = scene.render_shard(cam, w, h, spp, 0, R, fb.data_ptr(), flags=api.FLAG_TIME_KERNELS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
sys.stderr.flush()
print(f"{name} x{spp} R={R} wide={os.environ.get('RT_BVH_WIDE', '1')}: {dt*1e3:.2f} ms, k_paths {st['seconds_trace']*1e3:.2f} ms, "
      f"{st['camera_rays']/dt/1e6:.1f} Msamples/s, closest {st['closest_rays']} any {st['any_rays']} shade {st['shade_events']}", flush=True)
