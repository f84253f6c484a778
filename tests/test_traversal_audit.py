"""Traversal audit (CPU): where the reference's BVH walk and the product's BVH walk can differ, and who is right.

The image is defined by the triangle test (triangle.cuh:39-58).  A BVH only culls -- but the reference's slab test
(aabb_intersector.cuh:14-36: fp32, exact boxes, `entry <= exit`) is not conservative: about once in 10^7 rays it
rejects the box of a triangle that the triangle test would accept (first seen on the zero-thickness box of a light
triangle).  Which hits get lost depends on the reference's tree and rounding, so no other tree can reproduce them.

What is checked here, without a GPU:
  * rtcuda_amd/librt_hostcheck.so walks the PRODUCT's tree (rt_bvh.h: 2-wide records, boxes padded by 2 ulps, exit
    distance widened) on the CPU with the control flow and fp32 expressions of the HIP kernels;
  * every ray an oracle render traces (oracle ray log) is replayed through that walk;
  * wherever the product's walk and the reference's literal walk disagree, exhaustive search over all triangles
    decides -- and must side with the product;
  * the oracle's `watertight` mode (conservative box decisions + a tree-independent rule for hits at exactly equal t:
    the larger caller index wins, where the reference lets its tree order decide -- SURVEY Appendix A.10) agrees with
    the product's walk on EVERY ray, triangle index included, which is why the GPU parity tests can demand equal
    integer event totals against it at any frame size.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, default_camera

FLT_MAX = np.float32(3.4028234663852886e38)


class HostWalk:
    def __init__(self, arrays):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "rtcuda_amd", "csrc"), "../librt_hostcheck.so"],
                              stdout=subprocess.DEVNULL)
        self.lib = ctypes.CDLL(os.path.join(ROOT, "rtcuda_amd", "librt_hostcheck.so"))
        self.lib.rt_hostwalk_create.restype = ctypes.c_void_p
        self.lib.rt_hostwalk_create.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.lib.rt_hostwalk_destroy.argtypes = [ctypes.c_void_p]
        self.lib.rt_hostwalk_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6
        self.tris = np.ascontiguousarray(arrays.tris, np.float32)
        self.h = self.lib.rt_hostwalk_create(self.tris.ctypes.data, len(self.tris))
        assert self.h

    def closest(self, o, d, tmax=None):
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        n = len(o)
        tm = np.full(n, FLT_MAX, np.float32) if tmax is None else np.ascontiguousarray(tmax, np.float32)
        tri, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        assert self.lib.rt_hostwalk_trace(self.h, 0, n, o.ctypes.data, d.ctypes.data, tm.ctypes.data, None,
                                          tri.ctypes.data, t.ctypes.data) == 0
        return tri, t

    def any(self, o, d, tmax, excluded):
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        tm, ex = np.ascontiguousarray(tmax, np.float32), np.ascontiguousarray(excluded, np.int32)
        occ = np.zeros(len(o), np.int32)
        assert self.lib.rt_hostwalk_trace(self.h, 1, len(o), o.ctypes.data, d.ctypes.data, tm.ctypes.data,
                                          ex.ctypes.data, occ.ctypes.data, None) == 0
        return occ

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.rt_hostwalk_destroy(self.h)


@pytest.fixture(scope="module")
def walk_matte(bunny_matte):
    return HostWalk(bunny_matte)


# Path ray 1 836 499 of the matte 256 x 256 x 40 frame (seed 1), bit patterns of o and d: the reference's walk
# returns wall triangle 69458 at t = 0.4625543; the light triangle 69462 is hit at t = 0.4602134.
KNOWN_MISS_O = np.array([1052665855, 1062038174, 3212836608], np.uint32)
KNOWN_MISS_D = np.array([1048527110, 1054521025, 1063157561], np.uint32)


def _find_known_miss(oracle, bunny_matte):
    return KNOWN_MISS_O.view(np.float32).reshape(1, 3), KNOWN_MISS_D.view(np.float32).reshape(1, 3)


def test_known_ray_the_reference_walk_loses(oracle, bunny_matte, walk_matte):
    sc = oracle.scene(bunny_matte)
    o, d = _find_known_miss(oracle, bunny_matte)
    tm = np.full(1, FLT_MAX, np.float32)
    lit = sc.trace_closest(o, d, tm)
    brute = sc.trace_closest_brute(o, d, tm)
    prod = walk_matte.closest(o, d)
    wt = sc.set_watertight(True).trace_closest(o, d, tm)
    sc.set_watertight(False)
    assert brute[0][0] == 69462 and lit[0][0] == 69458  # light triangle in front of the wall behind it
    assert brute[1][0] < lit[1][0]
    assert prod[0][0] == brute[0][0] and prod[1][0] == brute[1][0]  # the product's walk = exhaustive search
    assert wt[0][0] == brute[0][0] and wt[1][0] == brute[1][0]      # and so is the oracle's watertight mode


@pytest.mark.parametrize("variant,w,h,spp", [("matte", 160, 160, 8), ("full_bsdf", 128, 72, 12)])
def test_every_ray_of_a_render_replayed_through_the_product_walk(oracle, variant, w, h, spp):
    from rtcuda_amd import scenes
    arrays = scenes.cornell_bunny(variant)
    sc = oracle.scene(arrays)
    walk = HostWalk(arrays)
    cam = default_camera(oracle, w / h)
    threads = os.cpu_count() or 8
    # ---- literal reference walk
    oracle.raylog_enable(True)
    _, _, st = sc.render(cam, w, h, spp, threads=threads)
    log = oracle.raylog_fetch()
    oracle.raylog_enable(False)
    assert len(log["any_tmax"]) == st["sum_ah"] and len(log["closest_tri"]) > 0
    occ = walk.any(log["any_o"], log["any_d"], log["any_tmax"], log["any_excluded"])
    bad = np.where(occ != log["any_occluded"])[0]
    assert len(bad) <= 1e-5 * len(occ)
    if len(bad):  # exhaustive search sides with the product on every disagreement
        br = sc.trace_any_brute(log["any_o"][bad], log["any_d"][bad], log["any_tmax"][bad], log["any_excluded"][bad])
        assert np.array_equal((br >= 0).astype(np.int32), occ[bad])
    tri, t = walk.closest(log["closest_o"], log["closest_d"])
    hit_c = log["closest_tri"] >= 0
    bad = np.where(((tri >= 0) != hit_c) | (hit_c & (t != log["closest_t"])))[0]
    assert len(bad) <= 1e-5 * len(tri)
    if len(bad):
        b = sc.trace_closest_brute(log["closest_o"][bad], log["closest_d"][bad], np.full(len(bad), FLT_MAX, np.float32))
        assert np.array_equal(b[0] >= 0, tri[bad] >= 0) and np.array_equal(b[1][tri[bad] >= 0], t[bad][tri[bad] >= 0])
    # the same hit distance on another triangle is a tie (shared edge): allowed, and counted
    ties = int(((tri != log["closest_tri"]) & hit_c & (tri >= 0) & (t == log["closest_t"])).sum())
    assert ties <= 1e-4 * len(tri)
    # ---- watertight oracle: the product's walk agrees on every single ray
    sc.set_watertight(True)
    oracle.raylog_enable(True)
    _, _, st_w = sc.render(cam, w, h, spp, threads=threads)
    log = oracle.raylog_fetch()
    oracle.raylog_enable(False)
    occ = walk.any(log["any_o"], log["any_d"], log["any_tmax"], log["any_excluded"])
    assert np.array_equal(occ, log["any_occluded"])
    tri, t = walk.closest(log["closest_o"], log["closest_d"])
    hit_c = log["closest_tri"] >= 0
    assert np.array_equal(tri >= 0, hit_c) and np.array_equal(t[hit_c], log["closest_t"][hit_c])
    assert np.array_equal(tri, log["closest_tri"])  # ties included: both sides give them to the larger caller index
    for k in ("sum_mat", "sum_ah", "emission_adds", "ah_adds", "rr_draws"):
        assert abs(st_w[k] - st[k]) <= 8, k  # the two modes differ by a handful of rays at most
