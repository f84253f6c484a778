#!/usr/bin/env python3
"""Copy what tools/refresh_profiles.sh measured (gpurun_out/<tag>/) into profiles/ and rebuild
profiles/pmc_k_paths.json, from which bench.py takes the per-launch counters of its roofline.

usage: update_profiles.py <tag>          e.g. r02

HBM bytes per k_paths launch = (2 * FETCH_SIZE + WRITE_SIZE) KB, as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts half of streaming reads; uncalibrated for this gather / scratch pattern)."""
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")


def counters(path, kernel="k_paths"):
    pmc = json.load(open(path))
    key = [k for k in pmc if kernel in k][0]
    return key, {n: v["mean_per_dispatch"] for n, v in pmc[key].items()}


def build_id_of(d):
    """The build the counters under `d` were collected on: tools/profile_scene.sh records rt_build_id() next to them."""
    p = os.path.join(d, "build_id.txt")
    return open(p).read().strip() if os.path.exists(p) else None


table_path = os.path.join(dst, "pmc_k_paths.json")
table = json.load(open(table_path)) if os.path.exists(table_path) else {}
for scene in ("full_bsdf", "four_bunnies", "sixteen_lights", "matte"):
    d = os.path.join(src, scene)
    if not os.path.exists(os.path.join(d, "pmc_summary.json")):
        continue
    shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(dst, f"{tag}_{scene}_kernel_stats.csv"))
    shutil.copy(os.path.join(d, "pmc_summary.json"), os.path.join(dst, f"{tag}_{scene}_pmc.json"))
    kname, c = counters(os.path.join(d, "pmc_summary.json"))
    gui = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    entry = {"kernel": "k_paths", "kernel_instance": kname, "build_id": build_id_of(d),
             "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"],
             "SQ_THREAD_CYCLES_VALU": c["SQ_THREAD_CYCLES_VALU"], "SQ_INSTS_SALU": c["SQ_INSTS_SALU"],
             "SQ_INSTS_VMEM": c["SQ_INSTS_VMEM"], "SQ_INSTS_LDS": c["SQ_INSTS_LDS"],
             "SQ_WAVE_CYCLES": c["SQ_WAVE_CYCLES"], "SQ_WAIT_ANY": c["SQ_WAIT_ANY"], "SQ_WAIT_INST_ANY": c["SQ_WAIT_INST_ANY"],
             "kernel_cycles": gui, "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
             "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
             "l2_hit_rate": c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0),
             "note": f"rocprofv3 --pmc passes of `bench.py --scene {scene} --spp 256 --steps 1 --warmup 0 --no-cpu-baseline "
                     f"--no-kernel-timing` (tools/profile_scene.sh; profiles/{tag}_{scene}_pmc.json); HBM bytes = (2*FETCH_SIZE + "
                     "WRITE_SIZE) KB per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts half of streaming reads)"}
    if "TA_BUSY_avr" in c:  # (texture addresser: busy cycles per unit, mean and maximum over the units, in kernel cycles)
        entry["TA_BUSY_avr"] = c["TA_BUSY_avr"]
        entry["TA_BUSY_max"] = c.get("TA_BUSY_max")
        entry["ta_busy_frac"] = c["TA_BUSY_avr"] / gui
        entry["ta_busy_max_frac"] = c.get("TA_BUSY_max", 0.0) / gui
    table[f"{scene}_1920x1080x256_n1"] = entry
    lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
    print("%-15s %.1f ms | VALU wave-instr %.4g | lane util %.1f %% | issue %.1f %% of 1 per 2 clk per SIMD | active lane-ops %.4g | "
          "HBM %.1f GB | L2 hit %.1f %% | wait_any %.0f %% wait_inst %.0f %% of wave cycles | VMEM wave-instr %.4g | TA busy %s" % (
              scene, gui / 2.4e6, c["SQ_INSTS_VALU"], 100 * lane_util, 100 * 2 * c["SQ_INSTS_VALU"] / (1024 * gui),
              c["SQ_INSTS_VALU"] * 64 * lane_util, entry["hbm_bytes_per_launch"] / 1e9, 100 * entry["l2_hit_rate"],
              100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_VMEM"],
              ("%.0f %% (max %.0f %%)" % (100 * entry["ta_busy_frac"], 100 * entry["ta_busy_max_frac"])) if "ta_busy_frac" in entry else "n/a"))
# one rank's shard of a 2-, 4-, 8-GPU run (tools/refresh_profiles.sh: shard_breakdown.py R under the SQ counter passes; no
# FETCH / WRITE pass, so no HBM traffic): the entries an N-GPU bench line prices rank 0's k_paths with
for R in (2, 4, 8):
    dR = os.path.join(src, f"shard{R}")
    if not os.path.exists(os.path.join(dR, "pmc_summary.json")):
        continue
    kname, c = counters(os.path.join(dR, "pmc_summary.json"))
    bid = build_id_of(dR) or build_id_of(os.path.join(src, "full_bsdf"))
    table[f"full_bsdf_1920x1080x256_n{R}"] = {
        "kernel": "k_paths", "kernel_instance": kname, "build_id": bid,
        "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"],
        "SQ_THREAD_CYCLES_VALU": c["SQ_THREAD_CYCLES_VALU"], "SQ_INSTS_SALU": c["SQ_INSTS_SALU"],
        "SQ_WAVE_CYCLES": c["SQ_WAVE_CYCLES"], "SQ_WAIT_ANY": c["SQ_WAIT_ANY"], "SQ_WAIT_INST_ANY": c["SQ_WAIT_INST_ANY"],
        "kernel_cycles": c["GRBM_GUI_ACTIVE"] / 8.0, "hbm_bytes_per_launch": None,
        "note": f"rocprofv3 --pmc passes of `tools/shard_breakdown.py {R}` (slot shard 0 of {R} of the full-BSDF frame on one MI355X: what "
                f"every rank of a {R}-GPU run renders; tools/refresh_profiles.sh; profiles/{tag}_shard{R}_pmc.json); no HBM pass"}
    shutil.copy(os.path.join(dR, "pmc_summary.json"), os.path.join(dst, f"{tag}_shard{R}_pmc.json"))
json.dump(table, open(table_path, "w"), indent=1)
for name in ("bench_n1.json", "shard_rate.txt", "valu_calibration.json", "valu_calibration_pmc.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
# shard-rate curve -> what bench.py echoes when n_gpus > 1
rates, rates_ps = {}, {}
p = os.path.join(src, "shard_rate.txt")
if os.path.exists(p):
    for ln in open(p):
        if ln.startswith("shards "):
            tok = ln.replace(":", "").split()
            rates[int(tok[1])] = float(ln.split("->")[1].split()[0])
        if ln.startswith("per_sample shards "):
            tok = ln.replace(":", "").split()
            rates_ps[int(tok[2])] = float(ln.split("->")[1].split()[0])
if rates:
    base = rates.get(1)
    json.dump({"source": f"tools/shard_rate.py on one MI355X ({tag}): the rate of ONE rank's slot shard, before the 24.9 MB reduce",
               "per_gpu_Msamples_per_s": rates,
               "predicted_Msamples_per_s": {n: round(n * r, 1) for n, r in rates.items()},
               "predicted_speedup": {n: round(n * r / base, 3) for n, r in rates.items()} if base else None,
               "per_sample_rng_mode": {
                   "note": "RT_FLAG_RNG_PER_SAMPLE (bench.py --rng-mode per_sample): NOT the reference's random numbers; every rank "
                           "runs the full slot pool on spp / N samples of every pixel",
                   "per_gpu_Msamples_per_s": rates_ps,
                   "predicted_Msamples_per_s": {n: round(n * r, 1) for n, r in rates_ps.items()},
                   "predicted_speedup": ({n: round(n * r / rates_ps[1], 3) for n, r in rates_ps.items()} if rates_ps.get(1) else None)}},
              open(os.path.join(dst, "shard_rate_prediction.json"), "w"), indent=1)
