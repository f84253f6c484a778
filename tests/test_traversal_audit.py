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
    integer event totals against it at any frame size;
  * round 5 -- the DEFAULT kernels' decision procedure (rtcuda_amd.hip: ref_visible + the rare literal re-trace), as its
    CPU twin `rt_hostwalk_trace_verified`: the product's own walk, every hit checked against what the reference's walk can
    SEE.  The reference's box test ignores tmax, so visibility is a function of the ray alone; the boxes on a root-to-leaf
    path are nested exactly and fp32 rounding is monotone, so a triangle is visible iff its LEAF's box passes the
    reference's slab test (and the triangle's own box passing implies that).  The verified walk must return the literal
    walk's answer on EVERY ray -- lost hits and ties included -- and does, on > 4 * 10^7 rays of literal renders plus
    axis-aligned, negative-zero and wall-hugging rays.
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
        self.lib.rt_hostwalk_trace_verified.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 7
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

    def closest_verified(self, o, d, tmax=None):
        """The default kernels' procedure for a path ray -> (triangle, t, stats6): see rt_hostwalk_trace_verified."""
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        n = len(o)
        tm = np.full(n, FLT_MAX, np.float32) if tmax is None else np.ascontiguousarray(tmax, np.float32)
        tri, t, st = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(6, np.int64)
        assert self.lib.rt_hostwalk_trace_verified(self.h, 0, n, o.ctypes.data, d.ctypes.data, tm.ctypes.data, None,
                                                   tri.ctypes.data, t.ctypes.data, st.ctypes.data) == 0
        return tri, t, dict(zip(("rays", "own_box_failed", "lost_hits", "ties", "neg_zero_rays", "literal_retraces"), st.tolist()))

    def any_verified(self, o, d, tmax, excluded):
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        tm, ex = np.ascontiguousarray(tmax, np.float32), np.ascontiguousarray(excluded, np.int32)
        occ, st = np.zeros(len(o), np.int32), np.zeros(6, np.int64)
        assert self.lib.rt_hostwalk_trace_verified(self.h, 1, len(o), o.ctypes.data, d.ctypes.data, tm.ctypes.data,
                                                   ex.ctypes.data, occ.ctypes.data, None, st.ctypes.data) == 0
        return occ, dict(zip(("rays", "own_box_failed", "lost_hits", "ties", "neg_zero_rays", "literal_retraces"), st.tolist()))

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


@pytest.mark.parametrize("variant,w,h,spp,lost_any,lost_closest", [
    ("matte", 256, 256, 40, 0, 1),            # the frame with the known lost hit (path ray 1 836 499)
    ("sixteen_lights", 480, 270, 48, 7, 0),   # 7 shadow rays whose occluder the reference cannot see
    ("full_bsdf", 200, 120, 24, None, None),
])
def test_verified_walk_returns_the_literal_walks_answer_on_every_ray(oracle, variant, w, h, spp, lost_any, lost_closest):
    """The decision procedure of the DEFAULT kernels against the reference's literal walk, ray by ray: every path ray and
    every shadow ray of a literal oracle render.  No tolerance: the hit triangle (ties included), t bit for bit, the
    occlusion flag.  The plain product walk (RT_FLAG_WATERTIGHT) differs on exactly the rays the counters name."""
    from rtcuda_amd import scenes
    arrays = scenes.cornell_bunny(variant)
    sc = oracle.scene(arrays)
    walk = HostWalk(arrays)
    oracle.raylog_enable(True)
    sc.render(default_camera(oracle, w / h), w, h, spp, threads=os.cpu_count() or 8)
    log = oracle.raylog_fetch()
    oracle.raylog_enable(False)
    occ, st_a = walk.any_verified(log["any_o"], log["any_d"], log["any_tmax"], log["any_excluded"])
    assert np.array_equal(occ, log["any_occluded"])
    plain = walk.any(log["any_o"], log["any_d"], log["any_tmax"], log["any_excluded"])
    assert int((plain != log["any_occluded"]).sum()) <= st_a["lost_hits"]  # a lost occluder flips the shadow ray (unless another hides it too)
    assert st_a["literal_retraces"] == st_a["lost_hits"]                    # a shadow ray is re-traced only for an unseen occluder
    tri, t, st_c = walk.closest_verified(log["closest_o"], log["closest_d"])
    hit = log["closest_tri"] >= 0
    assert np.array_equal(tri, log["closest_tri"])
    assert np.array_equal(t[hit].view(np.uint32), log["closest_t"][hit].view(np.uint32))
    assert st_c["literal_retraces"] == st_c["lost_hits"] + st_c["ties"] <= 2e-6 * len(tri) + 4
    if lost_any is not None:
        assert (st_a["lost_hits"], st_c["lost_hits"]) == (lost_any, lost_closest)
    # the own box is (nearly) as good a witness as the leaf's box: the common case never leaves the registers
    assert st_a["own_box_failed"] + st_c["own_box_failed"] <= 1e-6 * (len(tri) + len(occ)) + 8


def test_verified_walk_on_rays_chosen_to_break_it(oracle, bunny_matte, walk_matte):
    """Rays along the axes (the FLT_EPSILON clamp of 1 / d), rays with NEGATIVE-zero components (the reference's octant says
    "positive", its 1 / d is negative: nearly every box fails -- ref_visible then tests every ancestor), rays that start on
    the walls and rays that graze the flat boxes of the axis-aligned triangles: closest hit and any hit, verified walk
    against the oracle's literal walk, every ray."""
    import raygen
    sc = oracle.scene(bunny_matte)
    rng = np.random.default_rng(77)
    ao, ad = raygen.axis_aligned_rays(20000, seed=3)
    nz_o, nz_d = raygen.axis_aligned_rays(20000, seed=4)
    nz_d = np.where(nz_d == 0, np.float32(-0.0), nz_d)
    wall_o = rng.uniform(0, 1, (40000, 3)).astype(np.float32)
    wall_o[:, 2] = -wall_o[:, 2]
    wall_o[np.arange(40000), rng.integers(0, 3, 40000)] = rng.choice(np.array([0.0, 1.0, -1.0, 0.999], np.float32), 40000)
    wall_d = rng.normal(size=(40000, 3))
    wall_d[:20000, 1] *= 1e-3  # grazing the floor / ceiling / light planes
    wall_d = (wall_d / np.linalg.norm(wall_d, axis=1, keepdims=True)).astype(np.float32)
    o, d = np.concatenate([ao, nz_o, wall_o]), np.concatenate([ad, nz_d, wall_d])
    tmax = np.full(len(o), FLT_MAX, np.float32)
    want = sc.trace_closest(o, d, tmax)
    tri, t, st = walk_matte.closest_verified(o, d)
    assert st["neg_zero_rays"] >= 20000
    assert np.array_equal(tri, want[0])
    hit = want[0] >= 0
    assert np.array_equal(t[hit].view(np.uint32), want[1][hit].view(np.uint32))
    assert hit.sum() > 10000
    tm = rng.uniform(0.05, 2.0, len(o)).astype(np.float32)
    excl = rng.integers(-1, bunny_matte.n_tris, len(o)).astype(np.int32)
    occ, _ = walk_matte.any_verified(o, d, tm, excl)
    assert np.array_equal(occ, sc.trace_any(o, d, tm, excl))


def test_slab_term_is_monotone_in_the_bound():
    """The second premise of ref_visible: an fp32 slab term inv * bound + scaled_origin -- two roundings, as the reference
    computes it (aabb_intersector.cuh:25-33) -- is a monotone function of the bound (non-decreasing for inv > 0, non-increasing
    for inv < 0), because each rounding is.  Not a proof (rounding to nearest is monotone by definition), a tripwire: 4 * 10^6
    random (inv, origin, bound, bound') with bounds as close as neighbouring floats, magnitudes from 1e-7 to 8e6."""
    rng = np.random.default_rng(2025)
    n = 4_000_000
    f = np.float32
    inv = (rng.choice([-1.0, 1.0], n) * np.exp(rng.uniform(np.log(1e-1), np.log(8.4e6), n))).astype(f)
    o = rng.uniform(-2, 2, n).astype(f)
    so = (-o * inv).astype(f)
    b1 = rng.uniform(-2, 2, n).astype(f)
    steps = rng.integers(0, 4, n)
    b2 = b1.copy()
    for _ in range(3):
        b2 = np.where(steps > 0, np.nextafter(b2, f(np.inf)), b2)
        steps = steps - 1
    far = rng.random(n) < 0.3
    b2 = np.where(far, np.maximum(b1, rng.uniform(-2, 2, n).astype(f)), b2)
    assert (b2 >= b1).all()
    t1 = ((inv * b1).astype(f) + so).astype(f)
    t2 = ((inv * b2).astype(f) + so).astype(f)
    pos = inv > 0
    assert (t2[pos] >= t1[pos]).all() and (t2[~pos] <= t1[~pos]).all()
    # the FMA form a CUDA build would use (one rounding) is monotone as well: the same argument covers it
    u1 = (inv.astype(np.float64) * b1 + so).astype(f)
    u2 = (inv.astype(np.float64) * b2 + so).astype(f)
    assert (u2[pos] >= u1[pos]).all() and (u2[~pos] <= u1[~pos]).all()
