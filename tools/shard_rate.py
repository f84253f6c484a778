#!/usr/bin/env python3
"""Per-GPU rate of one slot shard for shard counts 1, 2, 4, 8 (what each rank of an N-GPU run does).

Predicts multi-GPU scaling on a one-GPU box: N-GPU throughput ~= N x rate(shard_count = N)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rtcuda_amd import api, scenes

w, h, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene = api.Scene(scenes.cornell_bunny("full_bsdf"))
cam = api.make_camera(aspect=w / h)
fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
base = None
for R in (1, 2, 4, 8):
    fb.zero_()
    scene.render_shard(cam, w, h, spp, 0, R, fb.data_ptr())  # warm-up (RNG states for this shard)
    fb.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    st = scene.render_shard(cam, w, h, spp, 0, R, fb.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    rate = st["camera_rays"] / dt / 1e6
    base = base or rate
    print(f"shards {R}: {st['camera_rays']} samples in {dt*1e3:.1f} ms -> {rate:.1f} Msamples/s per GPU; "
          f"predicted {R}-GPU throughput {R*rate:.0f} Msamples/s = {R*rate/base:.2f}x")

# the same curve in the per-sample RNG mode (NOT the reference's random numbers): a rank runs the full slot pool on spp / R
base = None
for R in (1, 2, 4, 8):
    fb.zero_()
    scene.render_shard(cam, w, h, spp, 0, R, fb.data_ptr(), flags=api.FLAG_RNG_PER_SAMPLE)
    fb.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    st = scene.render_shard(cam, w, h, spp, 0, R, fb.data_ptr(), flags=api.FLAG_RNG_PER_SAMPLE)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    rate = st["camera_rays"] / dt / 1e6
    base = base or rate
    print(f"per_sample shards {R}: {st['camera_rays']} samples in {dt*1e3:.1f} ms -> {rate:.1f} Msamples/s per GPU; "
          f"predicted {R}-GPU throughput {R*rate:.0f} Msamples/s = {R*rate/base:.2f}x")
