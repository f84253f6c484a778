#!/usr/bin/env python3
"""bench.py -- Msamples/s of the render path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work: started WITHOUT a launcher and with N > 1, this process -- before it imports torch or touches HIP --
starts N fresh rank processes itself (rtcuda_amd/dist.py: self_launch), relays rank 0's line and exits non-zero if any
rank did.

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
                oracle: integer event totals and image RMS -- the default (timed) kernels AND RT_FLAG_REFERENCE_WALK
                against the oracle's LITERAL mode (the reference's own tree, slab test and tie rule), RT_FLAG_WATERTIGHT
                against its watertight mode; all three pairs equal
  reference_decisions  how often the default kernels' rare paths ran in the timed frame: hits the reference's walk loses,
                exact ties, literal re-traces (~2 rays in 10^7), and the frame time of the same frame under
                RT_FLAG_WATERTIGHT (what making the reference's decisions costs)
  cpu_baseline  the CPU oracle (a port of the reference's algorithm; the reference itself needs
                nvcc + cuRAND + CUB and cannot be built here) timed on a bounded sample
  multi_gpu     (N > 1) ranks the collective backend saw, every rank's camera rays and kernel time, the reduce timed
                apart, and the frame's event totals summed over the ranks against the committed oracle totals
  per_sample    the same frame in the per-sample RNG mode (NOT the reference's random numbers; labelled so)
  extra_configs BASELINE configs 3, 4 and 5 (matte x 1024 spp, four bunnies x 256, sixteen lights x 512), a few frames
                each, event totals against the committed oracle totals
  reference_walk_mode  the headline frame once under RT_FLAG_REFERENCE_WALK (every ray through the reference's own tree: the
                slow cross-check of the default kernels): its rate, and its event totals against the committed totals of
                the oracle's LITERAL mode
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
TOTAL_KEYS = ("camera_rays", "shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws", "closest_rays",
              "literal_retraces", "reference_lost_hits", "exact_ties")
# BASELINE.json configs 3, 5 and 4 (the headline is config 2 = configs[1]); scene definitions in SURVEY.md section 8d
EXTRA_CONFIGS = (("matte", 1024, "BASELINE configs[2]: bun_zipper.ply 1920x1080 1024 spp (the reference's all-matte scene)"),
                 ("sixteen_lights", 512, "BASELINE configs[4]: bun_zipper.ply + 16 area lights, 1920x1080 512 spp"),
                 ("four_bunnies", 256, "BASELINE configs[3]: 4 bunnies (277 816 triangles), 1920x1080 256 spp"))


def closest_ray_bytes(np_c: float, tt_c: float) -> float:
    return 53.0 + 64.0 * np_c + 72.0 * tt_c


def any_ray_bytes(np_a: float, tt_a: float) -> float:
    return 68.0 + 64.0 * np_a + 72.0 * tt_a


def golden_totals(scene: str, w: int, h: int, spp: int, column: str = "oracle_literal"):
    """The committed oracle event totals of a full BASELINE frame (tests/golden/full_size_event_totals.json) or None.
    `oracle_literal`: the reference's own decisions -- what the default kernels reproduce; `oracle_watertight`: RT_FLAG_WATERTIGHT."""
    path = os.path.join(ROOT, "tests", "golden", "full_size_event_totals.json")
    if not os.path.exists(path):
        return None
    for f in json.load(open(path))["frames"]:
        if (f["scene"], f["width"], f["height"], f["spp"]) == (scene, w, h, spp):
            return f.get(column)
    return None


def parse_args():
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
    ap.add_argument("--no-extras", action="store_true",
                    help="headline configuration only: no per_sample sub-record, no extra_configs (profiling passes use this so "
                         "that every k_paths dispatch of the process is the headline frame's)")
    ap.add_argument("--rng-mode", default="reference", choices=["reference", "per_sample"],
                    help="reference: the reference's per-slot XORWOW streams (default; the parity mode; N ranks = N slot shards).  "
                         "per_sample: NOT the reference's random numbers -- one stream per camera ray, so every rank runs the full "
                         "slot pool on spp / N samples of every pixel (statistically equivalent image; see include/rtcuda_amd.h)")
    args = ap.parse_args()
    if args.debug_flags != 0 and not args.allow_invalid:
        raise SystemExit("--debug-flags changes what the kernels do (e.g. 0x100 drops every framebuffer deposit): the "
                         "number would be invalid.  Pass --allow-invalid to run anyway; the JSON line is then marked.")
    return args


class Config:
    """One configuration (scene, spp, RNG mode) on this rank: the scene on the device, its raw-sum buffers, and the frame step
    bench.py times -- zero, this rank's shard, ONE sum-reduce, post-process on rank 0 (rtcuda_amd/dist.py: frame_step)."""

    def __init__(self, env, arrays, w, h, spp, max_bounces, flags, deterministic):
        torch, api = env["torch"], env["api"]
        self.env, self.w, self.h, self.spp, self.max_bounces, self.flags, self.det = env, w, h, spp, max_bounces, flags, deterministic
        t0 = time.perf_counter()
        self.scene = api.Scene(arrays)  # BVH build + upload
        self.scene_create_s = time.perf_counter() - t0
        self.cam = api.make_camera(aspect=w / h)
        self.fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
        self.fb_fixed = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda") if deterministic else None
        self.local_sum = self.fb_fixed if deterministic else self.fb
        self.last = {}
        self.rng_init_s = None
        self.ref_tree_s = None

    def render_local(self):
        e = self.env
        fn = self.scene.render_shard_fixed if self.det else self.scene.render_shard
        return fn(self.cam, self.w, self.h, self.spp, e["rank"], e["world"], self.local_sum.data_ptr(),
                  max_bounces=self.max_bounces, seed=1, flags=self.flags, stream=e["stream"])

    def post(self):
        api, e = self.env["api"], self.env
        if self.det:
            api.post_process_fixed(self.fb_fixed.data_ptr(), self.fb.data_ptr(), self.w * self.h, self.spp, stream=e["stream"])
        else:
            api.post_process(self.fb.data_ptr(), self.w * self.h, self.spp, stream=e["stream"])

    def step(self):
        st = self.env["rtdist"].frame_step(self.local_sum.zero_, self.render_local, self.local_sum, self.post, self.env["rank"])
        self.last = st
        if self.rng_init_s is None:
            self.rng_init_s = round(st["seconds_rng_init"], 6)  # (first frame: later frames reuse the cached states)
            self.ref_tree_s = round(st["seconds_reference_tree"], 4)  # (first frame: the reference's own tree, built once per scene)
        return st

    def timed(self, steps, warmup, per_sample):
        """W untimed + K timed steps (barrier + device sync on both sides, MAX over ranks).  Returns (seconds, sums of the
        timed steps' statistics, failure message or '')."""
        e = self.env
        world, rank = e["world"], e["rank"]
        w, h, spp = self.w, self.h, self.spp
        agg = {"seconds_trace": 0.0, "closest_rays": 0, "launches_trace": 0, "seconds_advance": 0.0, "seconds_render": 0.0,
               "any_rays": 0, "shade_events": 0, "camera_rays": 0}
        state = {"on": False, "msg": ""}
        expect = w * h * spp // world  # reference mode: every rank owns W / world slots; per-sample mode: spp / world samples

        def step():
            st = self.step()
            if state["on"]:
                slack = 0 if (world == 1 or per_sample) else (1 << 20) // world
                if abs(st["camera_rays"] - expect) > slack:
                    state["msg"] = f"rank {rank}: timed step traced {st['camera_rays']} camera rays, expected {expect}"
                for k in agg:
                    agg[k] += st[k]

        for _ in range(warmup):
            step()
        state["on"] = True
        elapsed = e["rtdist"].timed_frames(step, steps, 0, device_sync=e["torch"].cuda.synchronize)
        return elapsed, agg, state["msg"]

    def frame_totals(self):
        """Event totals of the LAST frame summed over the ranks (a checksum over every scheduling decision, random number and
        ray of the frame: tests/golden/full_size_event_totals.json holds the oracle's)."""
        vals = [int(self.last.get(k, 0)) for k in TOTAL_KEYS]
        return dict(zip(TOTAL_KEYS, self.env["rtdist"].sum_over_ranks(vals)))

    def close(self):
        self.scene.close()


def main():
    args = parse_args()
    from rtcuda_amd import dist as rtdist  # (numpy only at import: the parent of a self-launch never touches torch or HIP)
    if args.gpus > 1 and not rtdist.launched_by_torchrun():
        raise SystemExit(rtdist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the product and there is no CPU fallback")
    n_dev = max(torch.cuda.device_count(), 1)
    if world > n_dev and args.backend == "nccl":
        raise SystemExit(f"--gpus {world} but only {n_dev} GPU(s) are visible: RCCL needs one device per rank "
                         f"(a one-GPU rehearsal of the N-rank flow: --backend gloo)")
    dev_index = local_rank % n_dev  # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)

    from rtcuda_amd import api, scenes

    w, h, spp = args.width, args.height, args.spp
    env = {"torch": torch, "api": api, "rtdist": rtdist, "rank": rank, "world": world,
           "stream": torch.cuda.current_stream().cuda_stream}
    base_flags = (0 if args.no_kernel_timing else api.FLAG_TIME_KERNELS) | args.debug_flags
    per_sample_headline = args.rng_mode == "per_sample"
    t_one = time.perf_counter()
    arrays = scenes.cornell_bunny(args.scene)   # PLY parse + the driver's scene recipe (host)
    t_recipe = time.perf_counter() - t_one
    head = Config(env, arrays, w, h, spp, args.max_bounces,
                  base_flags | (api.FLAG_RNG_PER_SAMPLE if per_sample_headline else 0), args.deterministic)
    one_off = {"scene_recipe_s": round(t_recipe, 4), "scene_create_s": round(head.scene_create_s, 4), "rng_init_s": None,
               "note": "outside the timed region (SURVEY 8d): PLY parse + scene recipe, BVH build + upload, one-off XORWOW state init, "
                       "the reference's own tree (first frame, inside the warm-up)"}

    # ---- the headline: warm-up outside, then K timed steps between barrier + synchronize, MAX over ranks (the driver's contract)
    elapsed, agg, fail_msg = head.timed(args.steps, args.warmup, per_sample_headline)
    one_off["rng_init_s"] = head.rng_init_s
    one_off["reference_tree_s"] = head.ref_tree_s  # host: the reference's binary SAH tree (bvh.cuh:30-219) for the rare re-traces + leaf boxes
    any_failed = rtdist.agree_on_failure(bool(fail_msg))
    if any_failed:  # every rank exits, together and non-zero; rank 0 still prints a line that says so
        if rank == 0:
            print(json.dumps({"metric": "Msamples/s at 1920x1080, bun_zipper.ply", "value": None, "unit": "Msamples/s",
                              "n_gpus": world, "invalid": fail_msg or "a rank reported a work-count mismatch"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)
    if args.save_image and rank == 0:
        scenes.write_ppm(args.save_image, head.fb.view(h, w, 3).cpu().numpy())
    head_totals = head.frame_totals()   # (collective: every rank calls it)
    invalid = ""

    # ---- N > 1: what the collective backend saw, every rank's share, the reduce on its own
    multi = None
    if world > 1:
        ones = rtdist.sum_over_ranks([1])[0]
        per_rank = rtdist.gather_from_ranks([int(head.last["camera_rays"]), int(round(1e6 * agg["seconds_trace"] / max(agg["launches_trace"], 1))),
                                             dev_index])
        frame_copy = head.fb.clone() if rank == 0 else None  # (the reduce timing below overwrites rank 0's buffer)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            rtdist.reduce_raw_sums(head.local_sum)
        torch.cuda.synchronize()
        dist.barrier()
        reduce_ms = 1e3 * (time.perf_counter() - t0) / reps
        multi = {"backend": dist.get_backend(), "ranks_seen": int(ones),
                 "per_rank": [{"rank": r, "camera_rays_per_frame": v[0], "k_paths_us": v[1], "device": v[2]} for r, v in enumerate(per_rank)],
                 "reduce_ms": round(reduce_ms, 3),
                 "reduce_note": f"{reps} sum-reduces of the {head.local_sum.numel() * head.local_sum.element_size() / 1e6:.1f} MB raw-sum "
                                f"framebuffer to rank 0, timed apart from the frames (inside a frame the reduce follows the slowest rank)"}
        if ones != world:
            invalid = f"the collective backend saw {ones} ranks, expected {world}"

    # ---- the frame's event totals, summed over the ranks, against the committed oracle totals of this frame
    want = golden_totals(args.scene, w, h, spp) if (not per_sample_headline and args.max_bounces == 10) else None
    totals_rec = {"summed_over_ranks": head_totals, "golden": want,
                  "equal": (all(head_totals[k] == v for k, v in want.items()) and head_totals["camera_rays"] == w * h * spp) if want else None,
                  "golden_source": "tests/golden/full_size_event_totals.json, column oracle_literal: the oracle in its LITERAL mode (the "
                                   "reference's own tree, fp32 slab test and tie rule); the default kernels equal it exactly"}
    if totals_rec["equal"] is False and args.debug_flags == 0:
        invalid = "the frame's event totals differ from the committed oracle totals"

    # ---- what making the reference's decisions costs: the same frame under RT_FLAG_WATERTIGHT (the triangle-list definition:
    # no check of the hits), and how often the default kernels' rare paths ran in the timed frame
    decisions = None
    if not per_sample_headline and args.debug_flags == 0:
        n_rays = float(head_totals["closest_rays"] + head_totals["any_rays"])
        decisions = {"literal_retraces": head_totals["literal_retraces"], "reference_lost_hits": head_totals["reference_lost_hits"],
                     "exact_ties": head_totals["exact_ties"],
                     "literal_retrace_fraction_of_rays": head_totals["literal_retraces"] / max(n_rays, 1.0),
                     "what": "default kernels = the product's walk + a check of every hit against what the reference's walk can see "
                             "(ref_visible); a closest hit that is invisible to it, or tied at the final distance, is re-traced through "
                             "the reference's own tree"}
        if not args.no_extras:
            wt = Config(env, arrays, w, h, spp, args.max_bounces, base_flags | api.FLAG_WATERTIGHT, args.deterministic)
            wt_elapsed, wt_agg, wt_fail = wt.timed(2, 1, False)
            wt_totals = wt.frame_totals()
            wt_failed = rtdist.agree_on_failure(bool(wt_fail))
            wt_want = golden_totals(args.scene, w, h, spp, "oracle_watertight") if args.max_bounces == 10 else None
            wt_equal = (all(wt_totals[k] == v for k, v in wt_want.items()) and wt_totals["camera_rays"] == w * h * spp) if wt_want else None
            decisions["watertight_flag"] = {
                "what": "RT_FLAG_WATERTIGHT, 2 timed frames: the triangle-list definition (no hit lost to a box test, ties by caller "
                        "index) -- NOT the reference's image on ~1 path in 4e6",
                "ms_per_frame": None if wt_failed else round(1e3 * wt_elapsed / 2, 3),
                "event_totals_equal_committed_WATERTIGHT_oracle_totals": wt_equal}
            wt.close()
            if wt_equal is False:
                invalid = "RT_FLAG_WATERTIGHT: the frame's event totals differ from the committed watertight-oracle totals"

    # ---- the same frame in the OTHER RNG mode (per-sample streams: NOT the reference's random numbers), same steps
    sub_per_sample = None
    if not args.no_extras and not per_sample_headline and args.debug_flags == 0 and spp % world == 0:
        ps = Config(env, arrays, w, h, spp, args.max_bounces, base_flags | api.FLAG_RNG_PER_SAMPLE, args.deterministic)
        ps_elapsed, ps_agg, ps_fail = ps.timed(args.steps, args.warmup, True)
        ps_totals = ps.frame_totals()
        ps_failed = rtdist.agree_on_failure(bool(ps_fail))
        rates_ok = all(abs(ps_totals[k] - head_totals[k]) <= 0.005 * head_totals[k] for k in ("shade_events", "any_rays"))
        sub_per_sample = {
            "parity": "NOT a parity mode: one XORWOW stream per camera ray instead of the reference's per-slot streams "
                      "(RT_FLAG_RNG_PER_SAMPLE; statistically equivalent image, exactly partition-invariant sums)",
            "value": None if (ps_failed or not rates_ok) else round(float(w) * h * spp * args.steps / ps_elapsed / 1e6, 3),
            "unit": "Msamples/s", "ms_per_step": round(1e3 * ps_elapsed / max(args.steps, 1), 3), "steps": args.steps,
            "warmup": args.warmup, "totals_summed_over_ranks": ps_totals,
            "self_check": {"shade_events_and_any_rays_per_sample_within_0.5pct_of_reference_mode": rates_ok,
                           "camera_rays_exact": ps_totals["camera_rays"] == w * h * spp},
            "parallelism": f"every rank: all 2^20 slots, {spp // world} of the {spp} samples of every pixel" if world > 1 else "1 GPU"}
        ps.close()

    # ---- a per-sample HEADLINE (--rng-mode per_sample) is checked against one untimed reference-mode frame: the same path
    # statistics (a lost or duplicated chunk of camera-ray ids, or a wrong stream key, would move them) -- ADVICE r3
    if per_sample_headline and args.debug_flags == 0:
        ref = Config(env, arrays, w, h, spp, args.max_bounces, base_flags, args.deterministic)
        ref.step()
        ref_totals = ref.frame_totals()
        ref.close()
        if head_totals["camera_rays"] != w * h * spp or any(
                abs(head_totals[k] - ref_totals[k]) > 0.005 * ref_totals[k] for k in ("shade_events", "any_rays")):
            invalid = "per_sample frame: camera rays or per-sample event rates differ from the reference-mode frame's"

    # ---- BASELINE configs 3, 5, 4: a few frames each, totals against the committed oracle totals
    extras = None
    if not args.no_extras and not per_sample_headline and args.debug_flags == 0 and (w, h) == (1920, 1080):
        extras = []
        for scene_name, e_spp, label in EXTRA_CONFIGS:
            cfg = Config(env, scenes.cornell_bunny(scene_name), w, h, e_spp, 10, base_flags, args.deterministic)
            e_steps = 2
            e_elapsed, e_agg, e_fail = cfg.timed(e_steps, 1, False)
            e_totals = cfg.frame_totals()
            e_failed = rtdist.agree_on_failure(bool(e_fail))
            e_want = golden_totals(scene_name, w, h, e_spp)
            e_equal = (all(e_totals[k] == v for k, v in e_want.items()) and e_totals["camera_rays"] == w * h * e_spp) if e_want else None
            extras.append({"config": label, "scene": scene_name, "spp": e_spp, "steps": e_steps, "warmup": 1,
                           "value": None if (e_failed or e_equal is False) else round(float(w) * h * e_spp * e_steps / e_elapsed / 1e6, 3),
                           "unit": "Msamples/s", "ms_per_frame": round(1e3 * e_elapsed / e_steps, 3),
                           "k_paths_ms_rank0": round(1e3 * e_agg["seconds_trace"] / max(e_agg["launches_trace"], 1), 3),
                           "event_totals_equal_committed_LITERAL_oracle_totals": e_equal,
                           "literal_retrace_fraction_of_rays": e_totals["literal_retraces"] / max(float(e_totals["closest_rays"] + e_totals["any_rays"]), 1.0),
                           "totals_summed_over_ranks": e_totals})
            cfg.close()
            if scene_name == "sixteen_lights":
                # the one BASELINE scene on which the triangle-list definition (RT_FLAG_WATERTIGHT; the timed kernels until round 4)
                # is NOT within 1e-4 RMS of the reference's literal walk (its slab test drops hits on the flat light boxes: DESIGN
                # section 3): the same frame once through the reference's own tree, as a cross-check of the timed frames above
                cfg = Config(env, scenes.cornell_bunny(scene_name), w, h, e_spp, 10, base_flags | api.FLAG_REFERENCE_WALK, args.deterministic)
                l_elapsed, _, l_fail = cfg.timed(1, 0, False)
                l_totals = cfg.frame_totals()
                l_failed = rtdist.agree_on_failure(bool(l_fail))
                l_want = golden_totals(scene_name, w, h, e_spp, "oracle_literal")
                l_equal = (all(l_totals[k] == v for k, v in l_want.items()) and l_totals["camera_rays"] == w * h * e_spp) if l_want else None
                extras[-1]["reference_walk_mode"] = {
                    "what": "RT_FLAG_REFERENCE_WALK, one frame: every ray through the reference's own tree -- the slow way to the totals the "
                            "timed (default) kernels produced above",
                    "value": None if (l_failed or l_equal is False) else round(float(w) * h * e_spp / l_elapsed / 1e6, 3),
                    "unit": "Msamples/s", "ms_per_frame": round(1e3 * l_elapsed, 3),
                    "event_totals_equal_committed_LITERAL_oracle_totals": l_equal,
                    "watertight_flag_vs_literal_walk": "RMS 3.2e-4, 7 678 pixels over 1e-4, -1 519 NEE deposits of 1.95e9 shadow rays "
                                                       "(profiles/r03_full_size_parity_sixteen_lights_512spp.json: what the timed kernels were "
                                                       "until round 4); the default kernels now lose those 1 285 occluders as the reference does"}
                cfg.close()

    # ---- the headline frame ONCE under RT_FLAG_REFERENCE_WALK (every ray through the reference's own tree, box test, order
    # and tie rule): its event totals against the committed totals of the oracle's LITERAL mode -- the slow cross-check of what
    # the timed kernels produce.  Never the timed kernels.
    ref_walk = None
    if not args.no_extras and not per_sample_headline and args.debug_flags == 0:
        cfg = Config(env, arrays, w, h, spp, args.max_bounces, base_flags | api.FLAG_REFERENCE_WALK, args.deterministic)
        r_elapsed, r_agg, r_fail = cfg.timed(1, 0, False)
        r_totals = cfg.frame_totals()
        r_failed = rtdist.agree_on_failure(bool(r_fail))
        r_want = golden_totals(args.scene, w, h, spp, "oracle_literal") if args.max_bounces == 10 else None
        r_equal = (all(r_totals[k] == v for k, v in r_want.items()) and r_totals["camera_rays"] == w * h * spp) if r_want else None
        ref_walk = {"what": "RT_FLAG_REFERENCE_WALK, one frame: the literal walk of the reference's own tree, cross-check mode: NOT the "
                            "timed kernels (which reach the same totals on their own walk)",
                    "value": None if (r_failed or r_equal is False) else round(float(w) * h * spp / r_elapsed / 1e6, 3),
                    "unit": "Msamples/s", "ms_per_frame": round(1e3 * r_elapsed, 3),
                    "event_totals_equal_committed_LITERAL_oracle_totals": r_equal, "totals_summed_over_ranks": r_totals}
        cfg.close()
        if r_equal is False:
            invalid = "RT_FLAG_REFERENCE_WALK: the frame's event totals differ from the committed literal-oracle totals"

    # ---- N > 1: the SAME frame through the in-process multi-device entry point (rt_render_multi: one host thread per GPU
    # in ONE process, peer copies to devices[0]) -- on rank 0, while the other ranks wait at the final barrier.  This is the
    # path a C++ driver reaches with render(..., devices) / RTCUDA_DEVICES; here it runs on the node's real GPUs.
    in_process = None
    if world > 1 and not args.no_extras and not per_sample_headline and args.debug_flags == 0 and rank == 0:
        import subprocess
        devices = [k % n_dev for k in range(world)]
        what = ("rt_render_multi over the same devices, in a CHILD process of rank 0 (the other ranks idle at a barrier; a failure in "
                "there costs this record, not the line): host buffers in and out, so the wall time includes allocation, the 24.9 MB "
                "copy back and -- first call -- the scene replicas")
        try:
            child_env = {k: v for k, v in os.environ.items()
                         if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multi_device_probe.py"), args.scene, str(w), str(h), str(spp),
                                str(args.max_bounces), ",".join(str(d) for d in devices)], capture_output=True, text=True,
                               timeout=240, env=child_env)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                in_process = {"what": what, "devices": devices, "error": f"exit code {p.returncode}: {p.stderr[-400:]}"}
            else:
                rec = json.loads(lines[-1])
                ref_mean = float(torch.nanmean(frame_copy.double()).item())
                in_process = dict(rec, what=what,
                                  event_totals_equal_the_rank_sharded_frame=all(rec["totals"][k] == head_totals[k] for k in rec["totals"]),
                                  image_mean_of_the_rank_sharded_frame=ref_mean)
        except Exception as e:  # noqa: BLE001 -- reported in the line, never fatal for it (incl. the child's timeout)
            in_process = {"what": what, "devices": devices, "error": str(e)}

    if rank == 0:
        samples = float(w) * h * spp * args.steps
        value = samples / elapsed / 1e6
        metric = "Msamples/s at 1920x1080, bun_zipper.ply" if (w, h) == (1920, 1080) else f"Msamples/s at {w}x{h}"
        if per_sample_headline:
            metric += " [per_sample RNG: NOT the reference's samples]"
        out = {
            "metric": metric,
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
            invalid = "debug_flags != 0: work was skipped inside the timed region"
        if per_sample_headline:
            out["scaling_mode"] = ("per_sample RNG streams: NOT the reference's image sample for sample (statistically equivalent; "
                                   "the shards' sums are exactly partition-invariant); every rank runs all 2^20 slots")
        out["event_totals"] = totals_rec
        if multi is not None:
            out["multi_gpu"] = multi
        out["reference_decisions"] = decisions
        out["per_sample"] = sub_per_sample
        out["extra_configs"] = extras
        out["reference_walk_mode"] = ref_walk
        if in_process is not None:
            out["multi_gpu"]["in_process_rt_render_multi"] = in_process
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
            # BASELINE config 1 IN FULL ("bun_zipper.ply, 256x256, 4 spp, diffuse-only, CPU reference path"): the
            # oracle on the whole frame, and the same frame on the HIP path beside it (one generation: the lockstep pipeline)
            if not args.no_extras:
                c1_arrays = scenes.cornell_bunny("matte")
                c1_cam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, 1.0)
                c1_img, _, c1_st = orc.scene(c1_arrays).render(c1_cam, 256, 256, 4, 10, 1, threads=cores)
                c1_gpu = api.Scene(c1_arrays)
                c1_gimg, c1_gst = c1_gpu.render(api.make_camera(aspect=1.0), 256, 256, 4)
                c1_gimg, c1_gst = c1_gpu.render(api.make_camera(aspect=1.0), 256, 256, 4)  # (second call: RNG states cached)
                c1_gpu.close()
                out["cpu_baseline"]["config_1_full"] = {
                    "workload": "BASELINE configs[0]: bun_zipper.ply 256x256, 4 spp, all matte (262 144 samples), whole frame",
                    "cpu_Msamples_per_s": round(262144 / c1_st["seconds_loop"] / 1e6, 4), "cpu_seconds_loop": round(c1_st["seconds_loop"], 3),
                    "cpu_iterations": int(c1_st["iterations"]),
                    "gpu_Msamples_per_s": round(262144 / max(c1_gst["seconds_render"], 1e-9) / 1e6, 1),
                    "gpu_ms": round(1e3 * c1_gst["seconds_render"], 3),
                    "events_equal": all(int(c1_gst[g]) == int(c1_st[o]) for g, o in (("shade_events", "sum_mat"), ("any_rays", "sum_ah"),
                                                                                     ("emission_adds", "emission_adds"), ("shadow_adds", "ah_adds"),
                                                                                     ("rr_draws", "rr_draws"))),
                    "rms_literal_oracle_vs_gpu": float(np.sqrt(np.mean((c1_img.astype(np.float64) - c1_gimg) ** 2)))}
            if ost["ch_rays"] > 0 and ost["ah_rays"] > 0:
                np_c = ost["ch_node_pairs"] / ost["ch_rays"]
                tt_c = ost["ch_tri_tests"] / ost["ch_rays"]
                np_a = ost["ah_node_pairs"] / ost["ah_rays"]
                tt_a = ost["ah_tri_tests"] / ost["ah_rays"]
                np_src = f"oracle statistics of the cpu_baseline sample ({args.cpu_spp} spp)"
            # ---- parity of THIS binary on THIS box: the sample frame on the GPU against the oracle.  Three pairs, each held
            # to equality: the default (timed) kernels and RT_FLAG_REFERENCE_WALK against the oracle's LITERAL mode -- the
            # reference's own tree, box test and tie rule: the run just timed above -- and RT_FLAG_WATERTIGHT against its
            # watertight mode.
            if not args.no_parity and args.debug_flags == 0 and not per_sample_headline:
                sc0 = head.scene
                gimg, gst = sc0.render(head.cam, w, h, args.cpu_spp, max_bounces=args.max_bounces, seed=1)
                timg, tst = sc0.render(head.cam, w, h, args.cpu_spp, max_bounces=args.max_bounces, seed=1, flags=api.FLAG_WATERTIGHT)
                wimg, _, wst = osc.set_watertight(True).render(ocam, w, h, args.cpu_spp, args.max_bounces, 1, threads=cores)
                pairs = (("shade_events", "sum_mat"), ("any_rays", "sum_ah"), ("emission_adds", "emission_adds"),
                         ("shadow_adds", "ah_adds"), ("rr_draws", "rr_draws"))

                def rms_of(a, b):
                    m = ~(np.isnan(a) | np.isnan(b))
                    return float(np.sqrt(np.mean((a[m].astype(np.float64) - b[m]) ** 2)))
                out["parity"] = {
                    "frame": f"{w}x{h}x{args.cpu_spp} ({w * h * args.cpu_spp / (1 << 20):.1f} generations: all but the last "
                             f"run in the persistent kernel)",
                    "oracle_mode": "LITERAL: the reference's own tree (bvh.cuh:30-219), fp32 slab test on exact boxes "
                                   "(aabb_intersector.cuh:14-36) and tie rule (triangle.cuh:49)",
                    "events_equal": all(gst[g] == ost[o] for g, o in pairs) and gst["camera_rays"] == w * h * args.cpu_spp,
                    "events": {g: [int(gst[g]), int(ost[o])] for g, o in pairs},
                    "rms": rms_of(gimg, oimg), "max_abs": float(np.nanmax(np.abs(gimg - oimg))),
                    "nan_pixels_gpu_oracle": [int(np.isnan(gimg).any(axis=2).sum()), int(np.isnan(oimg).any(axis=2).sum())],
                    "rare_paths": {k: int(gst[k]) for k in ("literal_retraces", "reference_lost_hits", "exact_ties")},
                    "watertight_flag": {
                        "what": "RT_FLAG_WATERTIGHT against the oracle's watertight mode (no hit lost to a box test, ties by caller "
                                "index): equal; against the literal oracle it differs by the rays the reference's fp32 slab test loses",
                        "events_equal": all(tst[g] == wst[o] for g, o in pairs), "rms": rms_of(timg, wimg),
                        "event_deltas_vs_literal": {g: int(tst[g]) - int(ost[o]) for g, o in pairs},
                        "rms_vs_literal": rms_of(timg, oimg),
                        "pixels_over_1e-4_vs_literal": int((np.nan_to_num(np.abs(timg - oimg)).max(axis=2) > 1e-4).sum())},
                    "tolerance": "north star: 1e-4 per-channel RMS; here and in the tests: equal event totals, RMS < 2e-6 (float atomics)"}
                if not out["parity"]["events_equal"] or out["parity"]["rms"] > 2e-6:
                    invalid = "parity check failed: the GPU frame differs from the LITERAL oracle"
                if not out["parity"]["watertight_flag"]["events_equal"] or out["parity"]["watertight_flag"]["rms"] > 2e-6:
                    invalid = "parity check failed: the RT_FLAG_WATERTIGHT frame differs from the watertight oracle"
                # ... and the mode that makes the reference's own decisions must reproduce the literal render exactly
                rimg, rst = sc0.render(head.cam, w, h, args.cpu_spp, max_bounces=args.max_bounces, seed=1, flags=api.FLAG_REFERENCE_WALK)
                out["parity"]["reference_walk_mode"] = {
                    "what": "RT_FLAG_REFERENCE_WALK (cross-check mode, not the timed kernels) against the literal oracle: every ray "
                            "through the reference's own tree, slab test, traversal order and tie rule on the GPU",
                    "events_equal": all(rst[g] == ost[o] for g, o in pairs) and rst["camera_rays"] == w * h * args.cpu_spp,
                    "events": {g: [int(rst[g]), int(ost[o])] for g, o in pairs},
                    "rms": rms_of(rimg, oimg), "max_abs": float(np.nanmax(np.abs(rimg - oimg))),
                    "ms_this_frame": round(1e3 * rst["seconds_render"], 2)}
                if not out["parity"]["reference_walk_mode"]["events_equal"] or out["parity"]["reference_walk_mode"]["rms"] > 2e-6:
                    invalid = "RT_FLAG_REFERENCE_WALK does not reproduce the literal oracle render"
                # the timed frames must carry the same per-sample work as the oracle's sample (statistical guard)
                for g, o in (("shade_events", "sum_mat"), ("any_rays", "sum_ah")):
                    r_gpu = agg[g] / (float(w) * h * spp * args.steps)
                    r_cpu = ost[o] / (float(w) * h * args.cpu_spp)
                    if abs(r_gpu - r_cpu) > 0.005 * r_cpu:
                        invalid = f"timed frames: {g} per sample {r_gpu:.4f} vs oracle sample {r_cpu:.4f}"
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
                # the second bound: the CU's texture addresser takes a load whose lanes read different records apart lane by
                # lane (~1 cycle per active lane: profiles/r05_gather_rate.txt); busy share of the kernel's cycles from the PMC pass
                if pmc.get("SQ_INSTS_VMEM"):
                    roof["vmem"] = {"wave_instructions_per_launch": pmc["SQ_INSTS_VMEM"],
                                    "per_ray": round(pmc["SQ_INSTS_VMEM"] * 64.0 / max(c_per + a_per, 1.0), 1),
                                    "ta_busy_frac": round(pmc["ta_busy_frac"], 4) if pmc.get("ta_busy_frac") else None,
                                    "ta_busy_max_frac": round(pmc["ta_busy_max_frac"], 4) if pmc.get("ta_busy_max_frac") else None,
                                    "note": "vector-memory wave-instructions (x 64 / rays = lane slots per ray, idle lanes included); TA_BUSY_avr / "
                                            "TA_BUSY_max over the kernel's cycles -- a saturated gather microbenchmark reads 0.84 / 0.96 "
                                            "(tools/lab/gather_rate.hip)"}
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
                # what the SIMDs can ISSUE for scalar code, measured in this run: independent v_mul_f32 chains at the kernel's
                # occupancy (rt_probe_issue; DESIGN 5.2: 2.9 cycles per wave64 instruction, and no scalar stream does better
                # except dependent sequences at 2.25) -- next to the kernel's own vector-instruction rate from the PMC pass
                sec, n_instr = api.probe_issue(2, 4, 20000)
                simple_ipc = 4.0 * n_instr / (sec * 2.4e9)  # wave-instructions per cycle per SIMD (nominal 2.4 GHz)
                roof["simple_stream_valu_per_cycle_per_simd_measured"] = round(simple_ipc, 3)
                if roof.get("valu_wave_instructions_per_launch"):
                    k_ipc = roof["valu_wave_instructions_per_launch"] / (1024.0 * avg_s * 2.4e9)
                    roof["kernel_valu_per_cycle_per_simd"] = round(k_ipc, 3)
                    roof["frac_of_simple_stream_issue"] = round(k_ipc / simple_ipc, 3)
            except Exception as e:  # noqa: BLE001 -- reported, never fatal for the bench line
                roof["calibration_error"] = str(e)
            roof["frac_note"] = ("frac = active lane-operations/s over the fp32 vector peak in lane-operations/s; "
                                 "valu_issue_frac counts every issued wave-instruction as 64 lanes (what the SIMDs spend "
                                 "issue slots on); the gap between the two is lane utilisation.  frac_of_simple_stream_issue = the "
                                 "kernel's vector instructions per cycle per SIMD over what a stream of independent v_mul_f32 "
                                 "issues at the same occupancy in this run (the spec-sheet 0.5 per cycle is not reached by any "
                                 "scalar stream on this chip: DESIGN 5.2).  HBM is not the bound: see hbm.frac")
            out["roofline"] = roof
        else:
            out["roofline"] = None
        out["one_off"] = one_off
        if invalid:
            out["invalid"] = invalid
            out["value"] = None  # an invalid line carries no number
        if world > 1:  # the 1-GPU shard measurements this scaling run can be held against
            pred = os.path.join(ROOT, "profiles", "shard_rate_prediction.json")
            if os.path.exists(pred):
                out["predicted_scaling"] = json.load(open(pred))
            out["scaling_modes"] = {
                "value": "THE PARITY MODE: the reference's per-slot XORWOW streams, slot-range shards -- the N-GPU image is the 1-GPU image "
                         "(event totals above: equal to the committed literal-oracle totals).  A slot's camera rays are sequential, so a "
                         "1/8 shard is 2 waves per SIMD whatever the kernel: predicted 4.1x at 8 GPUs, ceiling 4.2x (DESIGN section 6)",
                "per_sample.value": "NOT a parity mode (one stream per camera ray: a statistically equivalent image, exactly "
                                    "partition-invariant sums): every rank runs all 2^20 slots on spp / N samples -- the mode expected to "
                                    "meet the north star's >= 6x at 8 GPUs (predicted 7.3x)"}
        out["per_frame"] = {"closest_rays": agg["closest_rays"] // max(args.steps, 1),
                            "any_rays": agg["any_rays"] // max(args.steps, 1),
                            "shade_events": agg["shade_events"] // max(args.steps, 1),
                            "rounds": agg["launches_trace"] // max(args.steps, 1),
                            "shard_note": "per-frame counts are rank 0's shard" if world > 1 else "whole frame"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
