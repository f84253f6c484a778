#!/usr/bin/env python3
"""decide the wavefront split with measurements.

Runs rt_split_probe (include/rtcuda_amd_tools.h) on BASELINE configs[1]'s frame (C2: bun_zipper.ply, full BSDF set,
1920x1080x256): >= 64 M real rays of the frame's first generations are dumped from the round pipeline into dense
device arrays, then
  A  the trace kernel alone on them (device-resident, no PCIe), at 8 / 6 / 5 / 4 waves per SIMD -> Grays/s
  B  the shading code alone on the dumped shading records, 64 of 64 lanes shading, one material kind per launch
     -> Gshades/s; and gen() alone (round 0 of the pipeline: every slot generates)
  C  the frame's event totals (rays, shades, camera rays: the committed full-size parity record) divided by A and B:
     what separate trace and shade kernels would need for the frame if they ran back to back at those rates, with the
     queue traffic a split adds on top -- against the persistent kernel's measured frame time.
Writes one JSON object to stdout (kept as profiles/r03_split_probe.json).

usage (GPU box): python tools/split_probe.py [target_rays] [scene]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (loads the HIP runtime the library binds to)
from rtcuda_amd import api, scenes  # noqa: E402

w, h, spp = 1920, 1080, 256
target = int(sys.argv[1]) if len(sys.argv) > 1 else 64 << 20
variant = sys.argv[2] if len(sys.argv) > 2 else "full_bsdf"
# frame totals of the scene at 1920x1080x256 (profiles/r02_full_size_parity*.json: GPU == oracle)
FRAME = {"full_bsdf": {"samples": 530841600, "shades": 960164910, "any_rays": 652554779},
         "matte": {"samples": 530841600, "shades": 1010217083, "any_rays": 984060399}}


def run(wide: bool):
    if wide:
        os.environ.pop("RT_BVH_WIDE", None)
    else:
        os.environ["RTCUDA_EXPERIMENTAL"] = "1"
        os.environ["RT_BVH_WIDE"] = "0"
    sc = api.Scene(scenes.cornell_bunny(variant), library=api.tools_lib())
    r = api.split_probe(sc, api.make_camera(aspect=w / h), w, h, spp, target)
    sc.close()
    os.environ.pop("RT_BVH_WIDE", None)
    return r


out = {"frame": f"{variant} {w}x{h}x{spp}", "target_rays": target, "trees": {}}
for name, wide in (("4-wide (default)", True), ("2-wide", False)):
    r = run(wide)
    rays = r["closest_rays"] + r["any_rays"]
    e = {"raw": r, "rays_dumped": rays, "rounds": r["rounds"],
         "round_pipeline_per_round_us": {"k_advance": 1e6 * r["s_advance"] / max(r["rounds"] - 1, 1),
                                         "k_trace_pool": 1e6 * r["s_trace_pool"] / max(r["rounds"], 1)},
         "A_trace_only_Grays_per_s": {}, "B_shade_only_Gshades_per_s": {}}
    for wv in (8, 6, 5, 4):
        tc, ta = r[f"s_trace_closest_w{wv}"], r[f"s_trace_any_w{wv}"]
        e["A_trace_only_Grays_per_s"][f"{wv} waves/SIMD"] = {
            "closest": r["closest_rays"] / tc / 1e9 if tc else None, "any": r["any_rays"] / ta / 1e9 if ta else None,
            "both": rays / (tc + ta) / 1e9 if tc + ta else None, "blocks_per_cu": r[f"trace_blocks_per_cu_w{wv}"]}
    tot_s, tot_n = 0.0, 0.0
    for k in ("matte", "mirror", "glass"):
        if r[f"shades_{k}"]:
            e["B_shade_only_Gshades_per_s"][k] = r[f"shades_{k}"] / r[f"s_shade_{k}"] / 1e9
            tot_s += r[f"s_shade_{k}"]
            tot_n += r[f"shades_{k}"]
    e["B_shade_only_Gshades_per_s"]["all kinds, in the dump's proportions"] = tot_n / tot_s / 1e9 if tot_s else None
    e["B_gen_only_Ggen_per_s"] = (1 << 20) / r["s_advance_round0"] / 1e9 if r["s_advance_round0"] else None
    out["trees"][name] = e

# ---- C: the frame at those rates
if variant in FRAME:
    f = FRAME[variant]
    best = None
    for name, e in out["trees"].items():
        for wv, a in e["A_trace_only_Grays_per_s"].items():
            if a["both"] and (best is None or a["both"] > best[2]):
                best = (name, wv, a["both"], a["closest"], a["any"])
    e = out["trees"][best[0]]
    closest_rays = f["samples"] + f["shades"]  # one camera ray per sample + one path ray per shade
    t_trace = closest_rays / (best[3] * 1e9) + f["any_rays"] / (best[4] * 1e9)
    b_all = e["B_shade_only_Gshades_per_s"]["all kinds, in the dump's proportions"]
    t_shade = f["shades"] / (b_all * 1e9)
    t_gen = f["samples"] / (e["B_gen_only_Ggen_per_s"] * 1e9)
    rays = closest_rays + f["any_rays"]
    out["C_frame_at_those_rates"] = {
        "best_trace_build": {"tree": best[0], "waves": best[1], "Grays_per_s": best[2]},
        "rays": rays, "shades": f["shades"], "camera_rays": f["samples"],
        "ms_trace": 1e3 * t_trace, "ms_shade": 1e3 * t_shade, "ms_gen": 1e3 * t_gen,
        "ms_sum_back_to_back": 1e3 * (t_trace + t_shade + t_gen),
        "queue_traffic_GB": rays * 120 / 1e9,
        "note": "back-to-back sum of the three stages at their stand-alone rates (dense inputs, no tails, no queue "
                "management, no per-round launches); a split design adds ~120 B of queue traffic per ray on top. The "
                "persistent kernel's measured frame is in profiles/r03_bench_n1.json (r02: 136.5 ms)."}
print(json.dumps(out, indent=1))
