#!/usr/bin/env python3
"""bench.py -- Msamples/s of the render path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole frame of the headline configuration: bun_zipper.ply in the Cornell box,
full BSDF set (glass bunny, mirror back wall), 1920x1080, 256 spp, max_bounces 10, seed 1
(BASELINE.json configs[1]).  With N > 1 ranks the SAME frame is sharded by path slot
(rtcuda_amd/dist.py), each rank renders its slots into a raw-sum framebuffer that is already
resident in HBM, one RCCL sum-reduce brings the partial framebuffers to rank 0, and rank 0
post-processes: total work is fixed, so "scaling" is "strong".

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (k_paths: the whole asynchronous part of the frame in one persistent
                launch).  The kernel keeps rays, hit records and slot state in registers / LDS and gathers
                a 5.7 MB scene from L2, so HBM cannot bind it (measured traffic: `hbm`); what binds it is
                vector-ALU issue.  `bound` is therefore "valu": achieved = ACTIVE lane-operations per
                second (SQ_INSTS_VALU x 64 x lane utilisation of the committed PMC pass of this command,
                profiles/pmc_k_paths.json, / the launch duration measured live with HIP events), peak =
                256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T lane-op/s (the 157.3 TFLOP/s fp32 vector
                spec / 2), next to the rate a pure v_fma_f32 kernel sustains in the same run.  SURVEY 8d's
                algorithmic bytes at the REFERENCE's record sizes are kept as `reference_equivalent_GBs`.
  parity        (N = 1) the same scene at the CPU sample's spp rendered on the GPU and compared with the
                oracle: integer event totals and image RMS
  cpu_baseline  the CPU oracle (a port of the reference's algorithm; the reference itself needs
                nvcc + cuRAND + CUB and cannot be built here) timed on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes of the trace stage with the reference's record sizes:
#   closest-hit ray: 32 B ray in + 21 B hit out + 64 B per node PAIR visited + 72 B per triangle test
#   any-hit ray:     40 B ray in + 28 B accumulate + 64 B per node pair + 72 B per triangle test
# (72 B = 24-B Primitive + 48-B Triangle).  NP / TT are the REFERENCE traversal's per-ray averages
# for the workload; they are re-measured by the cpu_baseline leg (oracle statistics) when it runs
# and fall back to SURVEY.md Appendix C's values for the named config otherwise.
APPX_C = {  # scene: (NPc, TTc, NPa, TTa)
    "full_bsdf": (12.15, 4.09, 6.71, 5.83), "matte": (8.17, 3.32, 7.78, 6.10),
    "four_bunnies": (17.77, 4.84, 12.69, 6.78), "sixteen_lights": (8.31, 3.34, 10.87, 6.14)}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 at 2.4 GHz; peak fp32 vector 157.3 TFLOP/s = 2 flop x 78.6 T lane-op/s
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9


def closest_ray_bytes(np_c: float, tt_c: float) -> float:
    return 53.0 + 64.0 * np_c + 72.0 * tt_c


def any_ray_bytes(np_a: float, tt_a: float) -> float:
    return 68.0 + 64.0 * np_a + 72.0 * tt_a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--max-bounces", type=int, default=10)
    ap.add_argument("--scene", default="full_bsdf",
                    choices=["matte", "full_bsdf", "four_bunnies", "sixteen_lights"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=16, help="spp of the bounded CPU-baseline sample (~10-15 s of CPU work)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--save-image", default="")
    ap.add_argument("--deterministic", action="store_true",
                    help="accumulate in 64-bit fixed point (order-independent: the N-GPU image equals the 1-GPU image bit for bit)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--debug-flags", type=int, default=0, help="perf experiments only (results invalid)")
    ap.add_argument("--allow-invalid", action="store_true", help="accept --debug-flags != 0 (the line is marked invalid)")
    ap.add_argument("--no-parity", action="store_true", help="skip the GPU-vs-oracle parity block")
    ap.add_argument("--rng-mode", default="reference", choices=["reference", "per_sample"],
                    help="reference: the reference's per-slot XORWOW streams (default; the parity mode; N ranks = N slot shards).  "
                         "per_sample: NOT the reference's random numbers -- one stream per camera ray, so every rank runs the full "
                         "slot pool on spp / N samples of every pixel (statistically equivalent image; see include/rtcuda_amd.h)")
    args = ap.parse_args()
    if args.debug_flags != 0 and not args.allow_invalid:
        raise SystemExit("--debug-flags changes what the kernels do (e.g. 0x100 drops every framebuffer deposit): the "
                         "number would be invalid.  Pass --allow-invalid to run anyway; the JSON line is then marked.")

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the product and there is no CPU fallback")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)  # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)

    from rtcuda_amd import api, scenes, dist as rtdist

    w, h, spp = args.width, args.height, args.spp
    t_one = time.perf_counter()
    arrays = scenes.cornell_bunny(args.scene)   # PLY parse + the driver's scene recipe (host)
    t_recipe = time.perf_counter() - t_one
    scene = api.Scene(arrays)                   # BVH build + upload
    t_scene = time.perf_counter() - t_one - t_recipe
    one_off = {"scene_recipe_s": round(t_recipe, 4), "scene_create_s": round(t_scene, 4), "rng_init_s": None,
               "note": "outside the timed region (SURVEY 8d): PLY parse + scene recipe, BVH build + upload, one-off XORWOW state init"}
    cam = api.make_camera(aspect=w / h)
    fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    fb_fixed = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda") if args.deterministic else None
    stream = torch.cuda.current_stream().cuda_stream
    flags = (0 if args.no_kernel_timing else api.FLAG_TIME_KERNELS) | args.debug_flags
    if args.rng_mode == "per_sample":
        flags |= api.FLAG_RNG_PER_SAMPLE
    last_stats = {}

    def render_local():
        if args.deterministic:
            return scene.render_shard_fixed(cam, w, h, spp, rank, world, fb_fixed.data_ptr(), max_bounces=args.max_bounces,
                                            seed=1, flags=flags, stream=stream)
        return scene.render_shard(cam, w, h, spp, rank, world, fb.data_ptr(), max_bounces=args.max_bounces, seed=1,
                                  flags=flags, stream=stream)

    def post():
        if args.deterministic:
            api.post_process_fixed(fb_fixed.data_ptr(), fb.data_ptr(), w * h, spp, stream=stream)
        else:
            api.post_process(fb.data_ptr(), w * h, spp, stream=stream)

    local_sum = fb_fixed if args.deterministic else fb
    agg = {"seconds_trace": 0.0, "closest_rays": 0, "launches_trace": 0,
           "seconds_advance": 0.0, "seconds_render": 0.0, "any_rays": 0, "shade_events": 0}
    rays_per_rank = w * h * spp // world  # (every rank owns W / world slots; with spp | W / world exactly this many)
    timed = {"on": False}
    failure = {"msg": ""}  # a rank never leaves the collective sequence on its own: failures are agreed on after the loop

    def step():
        # one frame: zero -> this rank's slot shard -> ONE sum-reduce (RCCL) -> post-process on rank 0 (rtcuda_amd/dist.py)
        st = rtdist.frame_step(local_sum.zero_, render_local, local_sum, post, rank)
        last_stats.update(st)
        if one_off["rng_init_s"] is None:
            one_off["rng_init_s"] = round(st["seconds_rng_init"], 6)  # (first frame: later frames reuse the cached states)
        if timed["on"]:
            if world == 1 and st["camera_rays"] != w * h * spp:
                failure["msg"] = f"timed step traced {st['camera_rays']} camera rays, expected {w * h * spp}"
            if world > 1 and abs(st["camera_rays"] - rays_per_rank) > (1 << 20) // world:
                failure["msg"] = f"rank {rank}: timed step traced {st['camera_rays']} camera rays, expected ~{rays_per_rank}"
            for k in agg:
                agg[k] += st[k]

    # warm-up outside, then K timed steps between barrier + synchronize, MAX over ranks (the driver's contract)
    for _ in range(args.warmup):
        step()
    timed["on"] = True
    elapsed = rtdist.timed_frames(step, args.steps, 0, device_sync=torch.cuda.synchronize)
    any_failed = rtdist.agree_on_failure(bool(failure["msg"]))
    if any_failed:  # every rank exits, together and non-zero; rank 0 still prints a line that says so
        if rank == 0:
            print(json.dumps({"metric": "Msamples/s at 1920x1080, bun_zipper.ply", "value": None, "unit": "Msamples/s",
                              "n_gpus": world, "invalid": failure["msg"] or "a rank reported a work-count mismatch"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)

    if rank == 0:
        samples = float(w) * h * spp * args.steps
        value = samples / elapsed / 1e6
        out = {
            "metric": "Msamples/s at 1920x1080, bun_zipper.ply" if (w, h) == (1920, 1080) else f"Msamples/s at {w}x{h}",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (Cornell box + Stanford bunny recipe of the reference driver, seed 1)",
            "config": {"workload": f"bun_zipper.ply {w}x{h} {spp} spp, {args.scene} scene, max_bounces "
                                   f"{args.max_bounces} (BASELINE configs[1] when 1920x1080x256 full_bsdf)",
                       "scene": args.scene, "width": w, "height": h, "spp": spp, "max_bounces": args.max_bounces,
                       "parallelism": f"slot-shard x{world} + 1 RCCL reduce" if world > 1 else "1 GPU",
                       "accumulation": "int64 fixed point (order-independent)" if args.deterministic else "fp32 atomics",
                       "rng_mode": args.rng_mode,
                       "debug_flags": args.debug_flags},
            "ms_per_frame": round(1e3 * elapsed / max(args.steps, 1), 3),
        }
        if args.debug_flags != 0:
            out["invalid"] = "debug_flags != 0: work was skipped inside the timed region"
            out["value"] = None
        if args.rng_mode == "per_sample":
            out["scaling_mode"] = ("per_sample RNG streams: NOT the reference's image sample for sample (statistically equivalent; "
                                   "the shards' sums are exactly partition-invariant); every rank runs all 2^20 slots")
        # ---- CPU baseline (rank 0, N = 1 only): the oracle on a bounded sample of the same workload
        np_c, tt_c, np_a, tt_a = APPX_C[args.scene]
        np_src = "SURVEY.md Appendix C"
        out["parity"] = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle.oracle import Oracle, usable_cpus
            cores = usable_cpus()  # every CPU this process may use (all host cores; a box's cgroup quota caps it)
            orc = Oracle("pinned")
            osc = orc.scene(arrays)
            ocam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)
            oimg, _, ost = osc.render(ocam, w, h, args.cpu_spp, args.max_bounces, 1, threads=cores, collect_stats=True)
            cpu_v = w * h * args.cpu_spp / ost["seconds_loop"] / 1e6
            out["cpu_baseline"] = {
                "value": round(cpu_v, 4), "unit": "Msamples/s", "cores": cores, "host_logical_cpus": os.cpu_count(),
                "kind": "port",
                "sample": f"same scene and camera at {w}x{h}, {args.cpu_spp} spp ({w * h * args.cpu_spp} samples), "
                          f"render loop only ({ost['seconds_loop']:.1f} s; RNG init {ost['seconds_rng_init']:.2f} s "
                          f"reported apart), literal wavefront schedule, OpenMP over queue entries"}
            if ost["ch_rays"] > 0 and ost["ah_rays"] > 0:
                np_c = ost["ch_node_pairs"] / ost["ch_rays"]
                tt_c = ost["ch_tri_tests"] / ost["ch_rays"]
                np_a = ost["ah_node_pairs"] / ost["ah_rays"]
                tt_a = ost["ah_tri_tests"] / ost["ah_rays"]
                np_src = f"oracle statistics of the cpu_baseline sample ({args.cpu_spp} spp)"
            # ---- parity of THIS binary on THIS box: the sample frame on the GPU against the oracle -- its watertight
            # mode for the strict comparison (equal integer event totals), the literal reference walk (the run just
            # timed) beside it: that one loses about one accepted hit in 10^7 rays (tests/test_traversal_audit.py)
            if not args.no_parity and args.debug_flags == 0 and args.rng_mode == "reference":
                gimg, gst = scene.render(cam, w, h, args.cpu_spp, max_bounces=args.max_bounces, seed=1)
                wimg, _, wst = osc.set_watertight(True).render(ocam, w, h, args.cpu_spp, args.max_bounces, 1, threads=cores)
                pairs = (("shade_events", "sum_mat"), ("any_rays", "sum_ah"), ("emission_adds", "emission_adds"),
                         ("shadow_adds", "ah_adds"), ("rr_draws", "rr_draws"))

                def rms_of(a, b):
                    m = ~(np.isnan(a) | np.isnan(b))
                    return float(np.sqrt(np.mean((a[m].astype(np.float64) - b[m]) ** 2)))
                out["parity"] = {
                    "frame": f"{w}x{h}x{args.cpu_spp} ({w * h * args.cpu_spp / (1 << 20):.1f} generations: all but the last "
                             f"run in the persistent kernel)",
                    "events_equal": all(gst[g] == wst[o] for g, o in pairs) and gst["camera_rays"] == w * h * args.cpu_spp,
                    "events": {g: [int(gst[g]), int(wst[o])] for g, o in pairs},
                    "rms": rms_of(gimg, wimg), "max_abs": float(np.nanmax(np.abs(gimg - wimg))),
                    "nan_pixels_gpu_oracle": [int(np.isnan(gimg).any(axis=2).sum()), int(np.isnan(wimg).any(axis=2).sum())],
                    "vs_literal_reference_walk": {
                        "event_deltas": {g: int(gst[g]) - int(ost[o]) for g, o in pairs}, "rms": rms_of(gimg, oimg),
                        "pixels_over_1e-4": int((np.nan_to_num(np.abs(gimg - oimg)).max(axis=2) > 1e-4).sum())},
                    "tolerance": "north star: 1e-4 per-channel RMS; tests: equal event totals, RMS < 2e-6"}
                if not out["parity"]["events_equal"] or out["parity"]["rms"] > 1e-4:
                    out["invalid"] = "parity check failed: the GPU frame differs from the oracle"
                # ... and the LITERAL reference walk gates the line too: the product may differ from it only by the rays
                # the reference's fp32 slab test loses (about 1 in 10^7: tests/test_traversal_audit.py)
                lit = out["parity"]["vs_literal_reference_walk"]
                n_rays = float(gst["closest_rays"] + gst["any_rays"])
                ev_bound, px_bound = max(4, int(1e-6 * n_rays)), max(4, int(2e-7 * n_rays))
                lit["audited_bounds"] = {"abs_event_delta": ev_bound, "pixels_over_1e-4": px_bound, "rms": 1e-4}
                if (max(abs(v) for v in lit["event_deltas"].values()) > ev_bound or lit["pixels_over_1e-4"] > px_bound
                        or lit["rms"] > 1e-4):
                    out["invalid"] = "the GPU frame differs from the LITERAL reference walk by more than the audited bound"
                # the timed frames must carry the same per-sample work as the oracle's sample (statistical guard)
                for g, o in (("shade_events", "sum_mat"), ("any_rays", "sum_ah")):
                    r_gpu = agg[g] / (float(w) * h * spp * args.steps)
                    r_cpu = ost[o] / (float(w) * h * args.cpu_spp)
                    if abs(r_gpu - r_cpu) > 0.005 * r_cpu:
                        out["invalid"] = f"timed frames: {g} per sample {r_gpu:.4f} vs oracle sample {r_cpu:.4f}"
        else:
            out["cpu_baseline"] = None
        # ---- roofline of the dominant kernel.  Default pipeline: ONE k_paths launch per frame (init + mat + gen + ch +
        # ah of every slot as phases of a persistent kernel), so "per launch" = per frame.
        if agg["launches_trace"] > 0 and agg["seconds_trace"] > 0:
            launches = agg["launches_trace"]
            persistent = launches == args.steps
            kernel = "k_paths" if persistent else "k_trace<MODE_POOL>"
            c_per = agg["closest_rays"] / launches
            a_per = agg["any_rays"] / launches
            avg_s = agg["seconds_trace"] / launches
            # context only: SURVEY 8d's algorithmic bytes at the REFERENCE's record sizes,
            #   B = 153 g + 49 (max_bounces + 1) g + 430 m + closest_bytes c + any_bytes a
            # -- bytes the reference streams through HBM every iteration and this design keeps on chip
            ref_bytes = closest_ray_bytes(np_c, tt_c) * c_per + any_ray_bytes(np_a, tt_a) * a_per
            if persistent:
                g_per = float(w) * h * spp / world
                m_per = agg["shade_events"] / launches
                ref_bytes += (153.0 + 49.0 * (args.max_bounces + 1)) * g_per + 430.0 * m_per
            roof = {"kernel": kernel, "bound": "valu", "achieved": None, "peak": round(VALU_PEAK_LANE_OPS / 1e12, 2),
                    "unit": "Tlane-op/s", "frac": None, "traffic": None,
                    "avg_launch_us": round(avg_s * 1e6, 2), "launches": launches,
                    "closest_rays_per_launch": round(c_per, 1), "any_rays_per_launch": round(a_per, 1),
                    "grays_per_s": round((agg["closest_rays"] + agg["any_rays"]) / max(agg["seconds_trace"], 1e-12) / 1e9, 3),
                    "stage_share_of_render": {
                        "dominant": round(agg["seconds_trace"] / max(agg["seconds_render"], 1e-12), 4),
                        "advance": round(agg["seconds_advance"] / max(agg["seconds_render"], 1e-12), 4)},
                    "reference_equivalent_GBs": round(ref_bytes / avg_s / 1e9, 1),
                    "reference_equivalent_bytes_per_sample": round(ref_bytes / (float(w) * h * spp / world), 1) if persistent else None,
                    "np_tt_source": np_src}
            pmc_file = os.path.join(ROOT, "profiles", "pmc_k_paths.json")
            key = f"{args.scene}_{w}x{h}x{spp}_n{world}" + ("" if args.rng_mode == "reference" else "_" + args.rng_mode)  # (another build of the kernel)
            pmc = json.load(open(pmc_file)).get(key) if os.path.exists(pmc_file) else None
            roof["build_id"] = api.build_id()
            if pmc and pmc.get("kernel") == kernel and pmc.get("build_id") != roof["build_id"]:
                # counters of ANOTHER build (a kernel edit without a profile refresh): never priced against this launch time
                roof["pmc_source"] = (f"stale: profiles/pmc_k_paths.json[{key}] was collected on build {pmc.get('build_id')}, "
                                      f"the loaded library is {roof['build_id']}: achieved / frac / traffic are null")
            elif pmc and pmc.get("kernel") == kernel:
                lane_util = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
                lane_ops = pmc["SQ_INSTS_VALU"] * 64.0 * lane_util  # ACTIVE lane-operations of one launch
                roof["achieved"] = round(lane_ops / avg_s / 1e12, 3)
                roof["frac"] = round(lane_ops / avg_s / VALU_PEAK_LANE_OPS, 4)
                roof["lane_utilisation"] = round(lane_util, 4)
                roof["valu_wave_instructions_per_launch"] = pmc["SQ_INSTS_VALU"]
                roof["valu_issue_frac"] = round(pmc["SQ_INSTS_VALU"] * 64.0 / avg_s / VALU_PEAK_LANE_OPS, 4)
                roof["active_lane_ops_per_ray"] = round(lane_ops / max(c_per + a_per, 1.0), 1)
                roof["traffic"] = pmc.get("hbm_bytes_per_launch")
                if roof["traffic"]:
                    roof["hbm"] = {"achieved_GBs": round(roof["traffic"] / avg_s / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                                   "frac": round(roof["traffic"] / avg_s / 1e9 / HBM_PEAK_GBS, 4)}
                roof["pmc_source"] = pmc.get("note", "profiles/pmc_k_paths.json")
            else:
                roof["pmc_source"] = f"no committed PMC pass for {key}: achieved / frac / traffic are null"
            try:  # the two roofs measured on the box in the same run: vector-ALU issue and device copy bandwidth
                rate, _ = api.calibrate_valu(4, 20000)
                roof["peak_measured"] = round(rate / 1e12, 2)
                if roof["achieved"] is not None:
                    roof["frac_of_measured_peak"] = round(roof["achieved"] * 1e12 / rate, 4)
                roof["copy_bandwidth_measured_GBs"] = round(api.measure_copy_bandwidth(1 << 30, 5) / 1e9, 1)
            except Exception as e:  # noqa: BLE001 -- reported, never fatal for the bench line
                roof["calibration_error"] = str(e)
            roof["frac_note"] = ("frac = active lane-operations/s over the fp32 vector peak in lane-operations/s; "
                                 "valu_issue_frac counts every issued wave-instruction as 64 lanes (what the SIMDs spend "
                                 "issue slots on); the gap between the two is lane utilisation.  HBM is not the bound: see hbm.frac")
            out["roofline"] = roof
        else:
            out["roofline"] = None
        out["one_off"] = one_off
        if out.get("invalid"):
            out["value"] = None  # an invalid line carries no number
        if world > 1:  # the 1-GPU shard measurements this scaling run can be held against
            pred = os.path.join(ROOT, "profiles", "shard_rate_prediction.json")
            if os.path.exists(pred):
                out["predicted_scaling"] = json.load(open(pred))
        out["per_frame"] = {"closest_rays": agg["closest_rays"] // max(args.steps, 1),
                            "any_rays": agg["any_rays"] // max(args.steps, 1),
                            "shade_events": agg["shade_events"] // max(args.steps, 1),
                            "rounds": agg["launches_trace"] // max(args.steps, 1),
                            "shard_note": "per-frame counts are rank 0's shard" if world > 1 else "whole frame"}
        if args.save_image:
            img = fb.view(h, w, 3).cpu().numpy()
            scenes.write_ppm(args.save_image, img)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
