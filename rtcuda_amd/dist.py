"""Multi-GPU plumbing for the render path: one process per GPU, torch.distributed (RCCL over xGMI).

The path shards by PATH SLOT, not by pixel: rank r of R owns slots [r*W/R, (r+1)*W/R) and therefore
exactly the camera rays c with (c % W) in that range (render.cuh:254-261 ties camera ray c to the
c-th pending slot, and all W slots are pending together).  With spp | W/R that is a round-robin
deal of W/(R*spp)-pixel strips over the ranks, so every rank sees every part of the image.
There is no exchange inside the render; the only collective is ONE sum-reduce of the raw
framebuffers (width*height*3 fp32 = 24.9 MB at 1080p) to rank 0 after the last round.
"""
from __future__ import annotations

W = 1 << 20


def shard_range(rank: int, world: int):
    """Slot range [lo, hi) owned by ``rank`` (``world`` must divide W = 1048576)."""
    if world <= 0 or W % world != 0:
        raise ValueError("world size must divide 1048576")
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    n = W // world
    return rank * n, (rank + 1) * n


def owner_of_camera_ray(camera_ray_id: int, world: int) -> int:
    """Rank that renders camera ray ``camera_ray_id`` (pixel = id // spp, render.cuh:257)."""
    return (camera_ray_id % W) // (W // world)


def reduce_raw_sums(local_sum, dst: int = 0, group=None):
    """Sum-reduce the per-rank raw framebuffers to ``dst`` (in place on ``dst``).  Works on any
    torch.distributed backend: "nccl" (= RCCL) on the GPUs, "gloo" in the CPU tests."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(local_sum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return local_sum


def frame_step(zero, render_local, local_sum, post_process, rank: int, dst: int = 0, group=None):
    """One frame of the N-rank path, as bench.py runs it: zero the rank's raw-sum buffer, render the rank's slot shard
    into it, ONE sum-reduce to ``dst``, and ``post_process`` (sqrt(sum / spp), render.cuh:330-338) on ``dst`` only.
    The callables hide where the buffer lives (HBM with the HIP library, host memory with the CPU oracle in the tests).
    Returns whatever ``render_local`` returns (the shard's statistics)."""
    zero()
    stats = render_local()
    reduce_raw_sums(local_sum, dst=dst, group=group)
    if rank == dst:
        post_process()
    return stats


def agree_on_failure(failed: bool, group=None) -> bool:
    """True on EVERY rank if any rank failed (one MAX all-reduce of a flag).  A rank that raised on its own in the
    middle of a step would leave its peers blocked in the next collective for ever."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return bool(failed)
    t = torch.tensor([1 if failed else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


def _small_tensor(values, group=None):
    """An int64 tensor for a bookkeeping collective, on the device the backend wants (RCCL: the current GPU; gloo: host)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in values], dtype=torch.int64)
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        t = t.cuda()
    return t


def sum_over_ranks(values, group=None):
    """Element-wise sum of a short list of integers over all ranks (every rank gets the result; one all-reduce).  With the
    values [1] it counts the ranks the backend really connected."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return [int(v) for v in values]
    t = _small_tensor(values, group)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(x) for x in t.cpu().tolist()]


def gather_from_ranks(values, group=None):
    """Every rank's short list of integers, by rank (every rank gets all of them; one all-gather)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return [[int(v) for v in values]]
    t = _small_tensor(values, group)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [[int(x) for x in o.cpu().tolist()] for o in out]


def timed_frames(step, steps: int, warmup: int, device_sync=None, group=None) -> float:
    """bench.py's timing contract: ``warmup`` untimed steps, then exactly ``steps`` steps bracketed by a barrier and a
    device synchronisation on both sides; returns the MAX over ranks of the elapsed seconds (every rank gets it)."""
    import time

    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def fence():
        if multi:
            dist.barrier(group=group)
        if device_sync is not None:
            device_sync()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    return elapsed


def self_launch(script: str, argv, n_ranks: int, grace_s: float = 60.0, extra_env=None) -> int:
    """Start ``n_ranks`` fresh processes of ``script argv...`` -- one rank per GPU, the torch.distributed.run environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) set for each -- and wait for them.  Returns the exit code
    for the parent: 0 if every rank exited 0, otherwise the first non-zero code.

    This is how ``python bench.py --gpus N`` runs when no launcher started it.  The parent must not have touched the GPU
    (it imports neither torch nor the HIP library before calling this): the children are new processes, nothing is
    exec'ed over a process that holds the device.  The ranks inherit stdout / stderr, so rank 0's JSON line is the
    parent's output.  If a rank dies while its peers wait in a collective, the peers are given ``grace_s`` seconds and
    then terminated (the exact processes started here, by PID).
    """
    import os
    import socket
    import subprocess
    import sys
    import time

    if n_ranks < 1:
        raise ValueError("n_ranks must be positive")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "LOCAL_WORLD_SIZE": str(n_ranks)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver supports dmabuf IPC only: RCCL needs it)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    rc = 0
    first_failure = None
    pending = set(range(n_ranks))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                first_failure = time.monotonic()
        if pending and first_failure is not None and time.monotonic() - first_failure > grace_s:
            for r in pending:  # peers of a dead rank, most likely blocked in a collective
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        if pending:
            time.sleep(0.05)
    return rc if rc >= 0 else 1  # (a rank killed by a signal reports a negative code)


def launched_by_torchrun() -> bool:
    """True when this process already is one rank of a launched job (torch.distributed.run, or self_launch above)."""
    import os
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ
