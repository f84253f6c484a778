#!/usr/bin/env python3
"""The ORACLE's image of the six full BASELINE frames, as 64-bit hashes -> tests/golden/full_size_image_hashes.json.

Fixture generator (test infrastructure, like make_golden.py); CPU only (no GPU, no HIP library): for each frame and oracle mode (watertight / literal) one whole-frame render of
oracle/oracle.cpp with the product's fixed-point accumulation beside the float sums (render_literal, `fb_fixed`), hashed by
oracle.sums_hash.  The GPU suite renders the same frames with RT_FLAG_DETERMINISTIC (default kernels -> watertight hashes,
RT_FLAG_REFERENCE_WALK -> literal hashes) and compares hashes: full-size image parity, bit for bit, in 10 s of GPU time.
The event totals of every render are checked against the committed totals (tests/golden/full_size_event_totals.json) on
the way -- a mismatch aborts.  Minutes of CPU per frame and mode; results are written after every render, and frames
already in the file are skipped (the run can be resumed).

usage: python tests/golden/make_full_size_hashes.py [--threads N] [--only scene:spp[:mode]] ..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle.oracle import Oracle, sums_hash, usable_cpus  # noqa: E402
from rtcuda_amd import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "full_size_image_hashes.json")
TOTALS = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_event_totals.json")))["frames"]
PAIRS = (("shade_events", "sum_mat"), ("any_rays", "sum_ah"), ("emission_adds", "emission_adds"), ("shadow_adds", "ah_adds"),
         ("rr_draws", "rr_draws"))
threads = usable_cpus()
only = []
argv = sys.argv[1:]
while argv:
    a = argv.pop(0)
    if a == "--threads":
        threads = int(argv.pop(0))
    elif a == "--only":
        only.append(tuple(argv.pop(0).split(":")))
done = json.load(open(OUT)) if os.path.exists(OUT) else {
    "source": "tests/golden/make_full_size_hashes.py: oracle/oracle.cpp, pinned flavour, whole frames, seed 1, max_bounces 10; "
              "sums_sha256_64 = first 16 hex digits of SHA-256 over the little-endian int64 fixed-point sums (2^-30), row-major",
    "frames": []}
have = {(f["scene"], f["spp"], f["mode"]) for f in done["frames"]}
orc = Oracle("pinned")
w, h = 1920, 1080
cam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)
for tot in sorted(TOTALS, key=lambda f: f["spp"]):
    scene, spp = tot["scene"], tot["spp"]
    osc = None
    for mode in ("watertight", "literal"):
        if (scene, spp, mode) in have:
            continue
        if only and not any(o[0] == scene and int(o[1]) == spp and (len(o) < 3 or o[2] == mode) for o in only):
            continue
        if osc is None:
            osc = orc.scene(scenes.cornell_bunny(scene))
        osc.set_watertight(mode == "watertight")
        fixed = np.zeros((h, w, 3), np.int64)
        t = time.time()
        img, _, st = osc.render(cam, w, h, spp, threads=threads, fixed_out=fixed)
        events = {g: int(st[o]) for g, o in PAIRS}
        want = tot["oracle_" + mode]
        if events != want:
            raise SystemExit(f"{scene} x{spp} {mode}: event totals {events} differ from the committed {want}")
        rec = {"scene": scene, "width": w, "height": h, "spp": spp, "mode": mode, "events": events,
               "sums_sha256_64": sums_hash(fixed), "nan_pixels_float_image": int(np.isnan(img).any(axis=2).sum()),
               "oracle_seconds": round(time.time() - t, 1), "threads": threads}
        done["frames"].append(rec)
        json.dump(done, open(OUT, "w"), indent=1)
        print(rec, flush=True)
