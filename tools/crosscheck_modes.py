#!/usr/bin/env python3
"""Default kernels against RT_FLAG_REFERENCE_WALK on many frames, bit for bit (GPU only; a tool, its output is kept under
profiles/).  The default kernels run the product's own walk and check every hit against what the reference's walk can see;
RT_FLAG_REFERENCE_WALK sends every ray through the reference's own tree.  Both accumulate in fixed point
(rt_render_shard_fixed), so two frames are equal iff their 64-bit sums are: any ray on which the two disagree -- a hit the
check should have rejected, a tie it missed -- shows as a different sum.  The GPU suite holds fixed cases to the oracle; this
sweep varies seed, size, samples, bounce limit and scene far beyond them.

usage: crosscheck_modes.py [n_seeds]"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rtcuda_amd import api, scenes  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
KEYS = ("camera_rays", "shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws")
cases = [(640, 360, 16, 10), (333, 217, 37, 10), (1920, 1080, 4, 10), (256, 256, 64, 3), (800, 450, 24, 25)]
out = {"build_id": api.build_id(), "frames": 0, "rays": 0, "unequal": [], "rare_paths": {"literal_retraces": 0, "reference_lost_hits": 0, "exact_ties": 0},
       "what": "fixed-point sums and event totals of the default kernels == RT_FLAG_REFERENCE_WALK, frame by frame"}
t0 = time.time()
for variant in ("full_bsdf", "matte", "sixteen_lights", "four_bunnies"):
    sc = api.Scene(scenes.cornell_bunny(variant))
    for (w, h, spp, mb) in cases:
        cam = api.make_camera(aspect=w / h)
        for seed in range(1, n_seeds + 1):
            sums = []
            for flags in (0, api.FLAG_REFERENCE_WALK):
                buf = torch.zeros(3 * w * h, dtype=torch.int64, device="cuda")
                st = sc.render_shard_fixed(cam, w, h, spp, 0, 1, buf.data_ptr(), max_bounces=mb, seed=seed, flags=flags)
                torch.cuda.synchronize()
                sums.append((hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:16], {k: int(st[k]) for k in KEYS}, st))
            same = sums[0][0] == sums[1][0] and sums[0][1] == sums[1][1]
            out["frames"] += 1
            out["rays"] += int(sums[0][2]["closest_rays"]) + int(sums[0][2]["any_rays"])
            for k in out["rare_paths"]:
                out["rare_paths"][k] += int(sums[0][2][k])
            if not same:
                out["unequal"].append({"scene": variant, "case": [w, h, spp, mb], "seed": seed, "default": sums[0][:2], "reference_walk": sums[1][:2]})
    sc.close()
    print(variant, out["frames"], "frames", out["rays"], "rays", len(out["unequal"]), "unequal", f"{time.time() - t0:.0f} s", file=sys.stderr, flush=True)
out["seconds"] = round(time.time() - t0, 1)
print(json.dumps(out, indent=1))
