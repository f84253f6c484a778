#!/usr/bin/env python3
"""Collects the ORACLE's event totals of the six full BASELINE frames into tests/golden/full_size_event_totals.json.

The totals come from tools/full_size_parity.py, which runs oracle/oracle.cpp (both modes) on a whole BASELINE frame
beside the GPU render and writes profiles/<tag>_full_size_parity*.json (100 - 400 s of 16 cores per frame and mode: too
long for the test suite, so the suite holds the answers).  usage: python tests/golden/make_full_size_totals.py r03"""
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
keys = ("shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws")
out = {"source": f"profiles/{tag}_full_size_parity*.json (oracle side only)", "frames": []}
for path in sorted(glob.glob(os.path.join(root, "profiles", f"{tag}_full_size_parity*.json"))):
    j = json.load(open(path))
    scene, size = j["frame"].split()
    w, h, spp = (int(x) for x in size.split("x"))
    out["frames"].append({"scene": scene, "width": w, "height": h, "spp": spp, "samples": j["samples"],
                          "oracle_watertight": {k: j["oracle_watertight"][k] for k in keys},
                          "oracle_literal": {k: j["oracle_literal"][k] for k in keys},
                          "oracle_nan_pixels": j["vs_watertight"]["nan_pixels_gpu_oracle"][1]})
json.dump(out, open(os.path.join(root, "tests", "golden", "full_size_event_totals.json"), "w"), indent=1)
print(len(out["frames"]), "frames")
