"""world_size-2 gloo test of the N > 1 path (CPU): slot-range shards + one sum-reduce == the unsharded frame.

The per-rank renderer here is the CPU oracle (tests may use it as the checker's engine); what is
under test is the sharding rule and the reduce plumbing of rtcuda_amd/dist.py that bench.py uses
with the "nccl" (RCCL) backend on the GPUs.
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, spp, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from rtcuda_amd import dist as rtdist
    from rtcuda_amd import scenes
    orc = Oracle("pinned")
    sc = orc.scene(scenes.cornell_bunny("matte", bunny=False))
    cam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)
    lo, hi = rtdist.shard_range(rank, world)
    _, part, st = sc.render(cam, w, h, spp, slot_lo=lo, slot_hi=hi, threads=2)
    local = torch.from_numpy(part.copy())
    rtdist.reduce_raw_sums(local, dst=0)
    mats = torch.tensor([st["sum_mat"]], dtype=torch.int64)
    dist.reduce(mats, dst=0)
    if rank == 0:
        np.savez(out_path, total=local.numpy(), mats=mats.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _bench_worker(rank, world, port, w, h, spp, out_dir):
    """The N > 1 branch of bench.py -- rtdist.frame_step inside rtdist.timed_frames -- with the CPU oracle as the
    per-rank renderer and host tensors as the raw-sum buffers (gloo)."""
    sys.path.insert(0, ROOT)
    import time
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from rtcuda_amd import dist as rtdist
    from rtcuda_amd import scenes
    orc = Oracle("pinned")
    sc = orc.scene(scenes.cornell_bunny("matte", bunny=False))
    cam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)
    lo, hi = rtdist.shard_range(rank, world)
    local = torch.zeros(h, w, 3, dtype=torch.float32)
    calls = {"render": 0, "post": 0}
    rays = []

    def render_local():
        calls["render"] += 1
        if rank == 1:
            time.sleep(0.05)  # the slower rank: the reported time must be ITS time
        _, part, st = sc.render(cam, w, h, spp, slot_lo=lo, slot_hi=hi, threads=2)
        local.add_(torch.from_numpy(part))
        rays.append(st["sum_gen"])
        return st

    def post():
        calls["post"] += 1
        local.mul_(1.0 / spp).sqrt_()  # post_process_framebuffer (render.cuh:330-338)

    def step():
        rtdist.frame_step(local.zero_, render_local, local, post, rank)

    t0 = time.perf_counter()
    elapsed = rtdist.timed_frames(step, steps=2, warmup=1)
    wall = time.perf_counter() - t0
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=local.numpy(), elapsed=elapsed, wall=wall,
             renders=calls["render"], posts=calls["post"])
    dist.barrier()
    dist.destroy_process_group()


def test_bench_multi_rank_step_and_timing_contract(tmp_path, oracle):
    """bench.py with N > 1: every rank renders its slot shard, ONE sum-reduce, rank 0 alone post-processes; K timed
    steps after W warm-up steps, the MAX over ranks of the elapsed time on every rank."""
    import torch.multiprocessing as mp
    from rtcuda_amd import scenes
    w, h, spp = 32, 20, 8
    mp.spawn(_bench_worker, args=(2, _free_port(), w, h, spp, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["renders"]) == int(r1["renders"]) == 3          # 1 warm-up + 2 timed steps on both ranks
    assert int(r0["posts"]) == 3 and int(r1["posts"]) == 0         # post-process on rank 0 only
    assert float(r0["elapsed"]) == float(r1["elapsed"])            # all_reduce(MAX): the same number everywhere
    assert float(r0["elapsed"]) >= 2 * 0.05                        # ... and it is the slower rank's
    assert float(r0["elapsed"]) <= float(r0["wall"]) + 1e-3        # only the timed steps are inside
    sc = oracle.scene(scenes.cornell_bunny("matte", bunny=False))
    cam = oracle.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)
    full, _, _ = sc.render(cam, w, h, spp, threads=4)
    assert np.allclose(r0["image"], full, rtol=1e-5, atol=1e-6)    # rank 0 holds the whole post-processed frame
    assert full.sum() > 0


def test_two_rank_shards_reduce_to_the_full_frame(tmp_path, oracle):
    import torch.multiprocessing as mp
    from rtcuda_amd import scenes
    w, h, spp = 40, 24, 8
    out_path = str(tmp_path / "reduced.npz")
    mp.spawn(_worker, args=(2, _free_port(), w, h, spp, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    sc = oracle.scene(scenes.cornell_bunny("matte", bunny=False))
    cam = oracle.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, w / h)
    _, full, st = sc.render(cam, w, h, spp, threads=4)
    assert int(got["mats"][0]) == st["sum_mat"]
    assert np.allclose(got["total"], full, rtol=1e-5, atol=1e-6)
    assert full.sum() > 0


def _failure_worker(rank, world, port, out_dir):
    """One rank sees a work-count mismatch in a timed step: as in bench.py it only RECORDS it, finishes the collective
    sequence with its peers, and all ranks learn of it together (rtdist.agree_on_failure)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtcuda_amd import dist as rtdist
    local = torch.ones(8)
    failed = {"flag": False}

    def step():
        rtdist.frame_step(local.zero_, lambda: {}, local, lambda: None, rank)
        if rank == 1:
            failed["flag"] = True  # (bench.py: the camera-ray count of the step is off)

    rtdist.timed_frames(step, steps=2, warmup=0)
    agreed = rtdist.agree_on_failure(failed["flag"])
    none = rtdist.agree_on_failure(False)
    with open(os.path.join(out_dir, f"fail{rank}.txt"), "w") as fh:
        fh.write(f"{int(agreed)} {int(none)}")
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_rank_does_not_strand_its_peers(tmp_path):
    import torch.multiprocessing as mp
    from rtcuda_amd import dist as rtdist
    mp.spawn(_failure_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "fail0.txt").read_text() == "1 0"  # rank 0 did not fail itself, but knows
    assert (tmp_path / "fail1.txt").read_text() == "1 0"
    assert rtdist.agree_on_failure(True) is True and rtdist.agree_on_failure(False) is False  # single process


# ----------------------------------------------------------------------------- `python bench.py --gpus N` without a launcher
def _run_worker(*extra, timeout=240):
    import json
    import subprocess
    worker = os.path.join(ROOT, "tests", "selflaunch_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, worker] + list(extra), capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_self_launch_starts_its_own_ranks_and_relays_rank_zero(oracle):
    """What `python bench.py --gpus 2` does when the driver starts it WITHOUT torch.distributed.run: the parent (no torch, no
    HIP) starts the ranks itself (rtcuda_amd.dist.self_launch), rank 0's one JSON line is the parent's output, exit code 0.
    The ranks here render with the CPU oracle over gloo (tests/selflaunch_worker.py mirrors bench.py's N > 1 flow): both
    ranks are seen by the backend, each owns half the camera rays, the totals summed over the ranks equal the unsharded
    frame's, and rank 0 holds the whole post-processed image."""
    w, h, spp = 1024, 600, 1  # 614 400 camera rays: more than W / 2, so rank 1's slots serve some of them
    rc, lines, err = _run_worker("--ranks", "2", "--width", str(w), "--height", str(h), "--spp", str(spp), "--steps", "1")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1  # one line, from rank 0
    out = lines[0]
    assert out["n_ranks"] == 2 and out["ranks_seen"] == 2 and out["failed"] is False
    assert [r[1] for r in out["per_rank"]] == [0, 1]
    assert [r[0] for r in out["per_rank"]] == [1 << 19, w * h * spp - (1 << 19)]  # slots [0, W/2) and [W/2, W)
    assert out["totals"] == out["totals_unsharded"]
    assert out["image_max_abs_diff"] < 1e-5 and out["image_sum"] > 0


def test_self_launch_reports_a_dead_rank_instead_of_hanging(oracle):
    """A rank that dies before its first collective: its peer would wait in init_process_group / the reduce for ever.  The
    parent gives the peer a grace period, terminates it (by PID) and exits non-zero with the dead rank's code."""
    import time
    t0 = time.monotonic()
    rc, lines, _ = _run_worker("--ranks", "2", "--fail-rank", "1", "--grace", "3")
    assert rc == 7
    assert lines == []
    assert time.monotonic() - t0 < 120


def test_bench_py_starts_ranks_itself_and_fails_loudly_without_a_gpu():
    """bench.py itself, as the driver invokes it (`python3 bench.py --gpus 2 ...`, no launcher): it must get as far as its
    ranks -- each of which refuses to run without the HIP library's GPU (there is no CPU fallback) -- and exit non-zero."""
    import subprocess
    import torch
    if torch.cuda.is_available():  # (checked BEFORE anything is started: on a GPU node bench.py --gpus 2 would run a whole bench)
        pytest.skip("a GPU is present: the N > 1 flow is exercised by the GPU rehearsal instead")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert p.stderr.count("bench.py needs a GPU") == 2  # both ranks were started and said so
    assert "launch with torch.distributed.run" not in p.stderr
