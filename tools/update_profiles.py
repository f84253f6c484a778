#!/usr/bin/env python3
"""Copy what tools/refresh_profiles.sh measured (gpurun_out/r01/) into profiles/ and recompute
profiles/pmc_traffic.json (the `roofline.traffic` field of bench.py) from the PMC summary.

HBM bytes per k_paths launch = (2 * FETCH_SIZE + WRITE_SIZE) KB, as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts half of streaming reads; uncalibrated for this gather / scratch pattern)."""
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r01")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "r01_kernel_stats.csv"))
shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(dst, "r01_pmc.json"))
shutil.copy(os.path.join(src, "bench_n1.json"), os.path.join(dst, "bench_r01_n1.json"))
for extra in ("shard_rate.txt", "scenes.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, "r01_" + extra))
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
key = [k for k in pmc if "k_paths" in k][0]
c = {n: v["mean_per_dispatch"] for n, v in pmc[key].items()}
traffic = {
    "full_bsdf_1920x1080x256_n1": {
        "kernel": "k_paths",
        "fetch_size_kb": c["FETCH_SIZE"],
        "write_size_kb": c["WRITE_SIZE"],
        "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 0 "
                "--no-cpu-baseline --no-kernel-timing` (profiles/r01_pmc.json, tools/refresh_profiles.sh); bytes = "
                "(2*FETCH_SIZE + WRITE_SIZE) KB per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts half of streaming "
                "reads; uncalibrated for this gather/scratch pattern).  Writes are the framebuffer atomics (one 64-byte "
                "write each) and register spills; reads are spills and L2 misses of the BVH gather (5.7 MB scene, 4 MB "
                "L2 per XCD).",
    }
}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
gui = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
print("k_paths: %.1f ms @ %.2f GHz" % (gui / 2.4e6, 2.4))
print("VALU busy %.1f %%  (4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles))" % (400.0 * c["SQ_ACTIVE_INST_VALU"] / (1024 * gui)))
print("lane utilisation %.1f %%  (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU))" % (100.0 * c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])))
print("HBM traffic %.1f GB per launch; L2 hit rate %.1f %%" % (traffic["full_bsdf_1920x1080x256_n1"]["hbm_bytes_per_launch"] / 1e9,
                                                              100.0 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
print("instructions per launch: VALU %.3g SALU %.3g VMEM %.3g LDS %.3g" % (c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_VMEM"], c["SQ_INSTS_LDS"]))
