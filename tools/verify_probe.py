#!/usr/bin/env python3
"""The default kernels (the reference's decisions on the product's walk) on the GPU: full BASELINE frames against the committed
LITERAL-oracle hashes, the cost against RT_FLAG_WATERTIGHT, and how often the rare paths run.  One JSON object on stdout."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle.oracle import sums_hash  # noqa: E402
from rtcuda_amd import api, scenes  # noqa: E402

frames = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_image_hashes.json")))["frames"]
want = sys.argv[1:] or ["full_bsdf", "sixteen_lights"]
out = {"build_id": api.build_id(), "frames": []}
cache = {}
for f in frames:
    if f["scene"] not in want:
        continue
    w, h, spp = f["width"], f["height"], f["spp"]
    if f["scene"] not in cache:
        cache[f["scene"]] = api.Scene(scenes.cornell_bunny(f["scene"]))
    gpu = cache[f["scene"]]
    flags = 0 if f["mode"] == "literal" else api.FLAG_WATERTIGHT
    got = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    cam = api.make_camera(aspect=w / h)
    st = gpu.render_shard_fixed(cam, w, h, spp, 0, 1, got.data_ptr(), flags=flags)
    torch.cuda.synchronize()
    rec = {"scene": f["scene"], "spp": spp, "mode": f["mode"], "flags": flags,
           "events_equal": all(st[k] == v for k, v in f["events"].items()),
           "hash_equal": sums_hash(got.cpu().numpy()) == f["sums_sha256_64"],
           "literal_retraces": st["literal_retraces"], "reference_lost_hits": st["reference_lost_hits"], "exact_ties": st["exact_ties"]}
    # timing: float framebuffer, 3 frames
    fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    ms = []
    for _ in range(3):
        fb.zero_()
        s2 = gpu.render_shard(cam, w, h, spp, 0, 1, fb.data_ptr(), flags=flags)
        ms.append(round(s2["seconds_trace"] * 1e3, 2))
    rec["k_paths_ms"] = ms
    out["frames"].append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(out))
