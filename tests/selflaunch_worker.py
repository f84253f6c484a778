#!/usr/bin/env python3
"""A stand-in for bench.py's N > 1 flow that runs WITHOUT a GPU (test tooling for tests/test_dist_gloo.py).

Started plainly with `--ranks N` it does what `python bench.py --gpus N` does when no launcher started it: the parent --
which imports neither torch nor any renderer -- starts N rank processes through rtcuda_amd.dist.self_launch and exits with
their verdict.  A rank initialises the gloo backend, renders its slot shard with the CPU oracle (the per-rank renderer
here; on the GPUs it is the HIP library), runs bench.py's frame step (zero, shard, ONE sum-reduce, post-process on rank 0)
inside bench.py's timing contract, and rank 0 prints one JSON line with the multi-GPU bookkeeping bench.py reports.
`--fail-rank R`: rank R exits non-zero before its first collective (the parent must not hang and must exit non-zero).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--width", type=int, default=32)
    ap.add_argument("--height", type=int, default=20)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fail-rank", type=int, default=-1)
    ap.add_argument("--grace", type=float, default=5.0)
    args = ap.parse_args()
    from rtcuda_amd import dist as rtdist
    if args.ranks > 1 and not rtdist.launched_by_torchrun():
        assert "torch" not in sys.modules, "the parent of a self-launch must not have imported torch"
        raise SystemExit(rtdist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.ranks, grace_s=args.grace))

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if rank == args.fail_rank:
        raise SystemExit(7)
    import numpy as np
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    from oracle.oracle import Oracle
    from rtcuda_amd import scenes
    w, h, spp = args.width, args.height, args.spp
    orc = Oracle("pinned")
    sc = orc.scene(scenes.cornell_bunny("matte", bunny=False))
    cam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)
    lo, hi = rtdist.shard_range(rank, world)
    local = torch.zeros(h, w, 3, dtype=torch.float32)
    last = {}

    def render_local():
        _, part, st = sc.render(cam, w, h, spp, slot_lo=lo, slot_hi=hi, threads=2)
        local.add_(torch.from_numpy(part))
        last.update(st)
        return st

    def step():
        rtdist.frame_step(local.zero_, render_local, local, lambda: local.mul_(1.0 / spp).sqrt_(), rank)

    elapsed = rtdist.timed_frames(step, args.steps, args.warmup)
    # a rank's camera rays: the generation events of ITS slots (the oracle's shard render idles the other slots through gen)
    mine = sum(1 for c in range(w * h * spp) if lo <= c % rtdist.W < hi)
    totals = rtdist.sum_over_ranks([mine, last["sum_mat"], last["sum_ah"], last["ah_adds"], last["rr_draws"]])
    ranks_seen = rtdist.sum_over_ranks([1])[0]
    per_rank = rtdist.gather_from_ranks([mine, rank])
    failed = rtdist.agree_on_failure(False)
    if rank == 0:
        full, _, st_full = sc.render(cam, w, h, spp, threads=2)
        print(json.dumps({"n_ranks": world, "ranks_seen": ranks_seen, "elapsed_s": elapsed, "failed": failed,
                          "per_rank": per_rank, "totals": totals,
                          "totals_unsharded": [w * h * spp, st_full["sum_mat"], st_full["sum_ah"], st_full["ah_adds"], st_full["rr_draws"]],
                          "image_max_abs_diff": float(np.abs(local.numpy() - full).max()), "image_sum": float(full.sum())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
