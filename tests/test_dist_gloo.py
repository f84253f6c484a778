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
