#!/usr/bin/env python3
"""A/B timing of the frame under environment knobs, in ONE process on ONE box (boxes differ by a percent or two).

usage: ab_bench.py [--scene full_bsdf] [--spp 256] [--reps 3] [--shards 1] "K1=V1,K2=V2" "K1=V3" ...
       (an empty string "" = the defaults; knobs read at scene creation -- RT_BVH_* -- get a scene of their own)

Prints one line per setting: frame time (best and mean of `reps` after one warm-up), the kernel's own time, and the
event totals (which must not move: every knob here is result-invariant)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTCUDA_EXPERIMENTAL"] = "1"  # (the library reads its experiment knobs only under this gate)
import torch  # noqa: E402
from rtcuda_amd import api, scenes  # noqa: E402

args = sys.argv[1:]
opts = {"--scene": "full_bsdf", "--spp": "256", "--reps": "3", "--shards": "1"}
while args and args[0] in opts:
    opts[args[0]] = args[1]
    args = args[2:]
variant, spp, reps, shards = opts["--scene"], int(opts["--spp"]), int(opts["--reps"]), int(opts["--shards"])
w, h = 1920, 1080
arrays = scenes.cornell_bunny(variant)
cam = api.make_camera(aspect=w / h)
fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
base = None
for setting in (args or [""]):
    env = dict(kv.split("=", 1) for kv in setting.split(",") if kv)
    for k, v in env.items():
        os.environ[k] = v
    sc = api.Scene(arrays)
    times, st = [], None
    for r in range(reps + 1):
        fb.zero_()
        torch.cuda.synchronize()
        t = time.perf_counter()
        st = sc.render_shard(cam, w, h, spp, 0, shards, fb.data_ptr(), flags=api.FLAG_TIME_KERNELS)
        torch.cuda.synchronize()
        if r > 0:
            times.append(time.perf_counter() - t)
    digest = float(torch.nan_to_num(fb.double()).sum().item())  # (float atomics: equal sums to ~1e-9 relative between result-invariant knobs)
    sc.close()
    for k in env:
        del os.environ[k]
    best, mean = min(times) * 1e3, sum(times) / len(times) * 1e3
    base = base or best
    print(f"{setting or '(defaults)':40s} best {best:8.2f} ms  mean {mean:8.2f} ms  kernel {st['seconds_trace'] * 1e3:8.2f} ms  "
          f"x{base / best:.3f}  events {st['camera_rays']} {st['shade_events']} {st['any_rays']} {st['shadow_adds']} {st['rr_draws']}  image sum {digest:.6f}",
          flush=True)
