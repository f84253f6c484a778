"""ctypes front-end of the CPU oracle (oracle/oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this
module.  The product (rtcuda_amd/) never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

_u32p = ctypes.c_void_p
_fp = ctypes.c_void_p


def build(force: bool = False) -> None:
    """Compile both oracle flavours with the committed Makefile (g++)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _ptr(a):
    return None if a is None else a.ctypes.data


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota.  (A GPU box shows 256
    logical CPUs to a container that is allowed 16: 256 OpenMP threads on a 16-CPU quota run an oracle render ~10x slower.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:  # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return max(1, n)


class Oracle:
    """One loaded flavour of the oracle: ``"pinned"`` (default) or ``"libm"``."""

    def __init__(self, flavour: str = "pinned"):
        name = {"pinned": "liboracle.so", "libm": "liboracle_libm.so"}[flavour]
        path = os.path.join(_BUILD, name)
        if not os.path.exists(path):
            build()
        self.lib = L = ctypes.CDLL(path)
        self.flavour = flavour
        L.orc_build_info.restype = ctypes.c_char_p
        assert L.orc_build_info().decode() == flavour
        L.orc_xorwow_init.argtypes = [ctypes.c_uint64, ctypes.c_uint32, _u32p]
        L.orc_xorwow_init_range.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _u32p]
        L.orc_xorwow_draw.argtypes = [_u32p, ctypes.c_int, _u32p, _fp]
        L.orc_uniform_from_u32.argtypes = [ctypes.c_uint32]
        L.orc_uniform_from_u32.restype = ctypes.c_float
        L.orc_jump_row.argtypes = [ctypes.c_int, _u32p]
        L.orc_camera.argtypes = [_fp, _fp, _fp, ctypes.c_float, ctypes.c_float, _fp]
        L.orc_camera_get_ray.argtypes = [_fp, ctypes.c_float, ctypes.c_float, _fp]
        L.orc_offset_ray_origin.argtypes = [_fp, _fp, _fp]
        L.orc_power_heuristic.argtypes = [ctypes.c_float, ctypes.c_float]
        L.orc_power_heuristic.restype = ctypes.c_float
        L.orc_triangle.argtypes = [_fp, _fp, _fp]
        L.orc_triangle_intersect.argtypes = [_fp, _fp, _fp, ctypes.c_float, _fp]
        L.orc_triangle_intersect.restype = ctypes.c_int
        L.orc_sincos.argtypes = [ctypes.c_float, _fp, _fp]
        L.orc_pow5.argtypes = [ctypes.c_float]
        L.orc_pow5.restype = ctypes.c_float
        L.orc_sample_f.argtypes = [ctypes.c_void_p, _fp, _fp, _u32p, _fp]
        L.orc_scene_create.argtypes = [_fp, ctypes.c_int, _fp, _fp, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_int]
        L.orc_scene_create.restype = ctypes.c_void_p
        L.orc_scene_destroy.argtypes = [ctypes.c_void_p]
        L.orc_scene_bvh_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _fp]
        L.orc_scene_nodes.argtypes = [ctypes.c_void_p, _fp, _fp, _fp, _fp]
        L.orc_trace_closest.argtypes = [ctypes.c_void_p, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                        ctypes.c_int]
        L.orc_trace_closest_brute.argtypes = L.orc_trace_closest.argtypes
        L.orc_closest_ties.argtypes = [ctypes.c_void_p, _fp, _fp, ctypes.c_float, ctypes.c_float, _fp,
                                       ctypes.c_int]
        L.orc_closest_ties.restype = ctypes.c_int
        L.orc_trace_any.argtypes = [ctypes.c_void_p, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, ctypes.c_int]
        L.orc_trace_any_brute.argtypes = L.orc_trace_any.argtypes
        L.orc_scene_set_watertight.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_raylog_enable.argtypes = [ctypes.c_int]
        L.orc_raylog_counts.argtypes = [ctypes.c_void_p]
        L.orc_raylog_fetch_any.argtypes = [_fp, _fp]
        L.orc_raylog_fetch_closest.argtypes = [_fp, _fp, _fp]
        L.orc_render.argtypes = [ctypes.c_void_p, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 _fp, _fp, _fp, _fp, ctypes.c_int]

    # ------------------------------------------------------------------ RNG
    def xorwow_init(self, seed: int, subsequence: int) -> np.ndarray:
        st = np.zeros(6, np.uint32)
        self.lib.orc_xorwow_init(seed, subsequence, _ptr(st))
        return st

    def xorwow_init_range(self, seed: int, first: int, count: int) -> np.ndarray:
        st = np.zeros((count, 6), np.uint32)
        self.lib.orc_xorwow_init_range(seed, first, count, _ptr(st))
        return st

    def xorwow_draw(self, state: np.ndarray, n: int):
        raw = np.zeros(n, np.uint32)
        uni = np.zeros(n, np.float32)
        self.lib.orc_xorwow_draw(_ptr(state), n, _ptr(raw), _ptr(uni))
        return raw, uni

    def jump_rows(self) -> np.ndarray:
        """The 160 x 5-word GF(2) matrix of the 2^67-draw jump (row b = image of basis bit b)."""
        rows = np.zeros((160, 5), np.uint32)
        for b in range(160):
            self.lib.orc_jump_row(b, _ptr(rows[b]))
        return rows

    # ------------------------------------------------------------------ unit functions
    def camera(self, lookfrom, lookat, up, vfov: float, aspect: float) -> np.ndarray:
        out = np.zeros(12, np.float32)
        a, b, c = (np.asarray(v, np.float32) for v in (lookfrom, lookat, up))
        self.lib.orc_camera(_ptr(a), _ptr(b), _ptr(c), vfov, aspect, _ptr(out))
        return out

    def camera_get_ray(self, cam12, x: float, y: float) -> np.ndarray:
        out = np.zeros(6, np.float32)
        cam12 = np.ascontiguousarray(cam12, np.float32)
        self.lib.orc_camera_get_ray(_ptr(cam12), x, y, _ptr(out))
        return out

    def offset_ray_origin(self, p, n) -> np.ndarray:
        out = np.zeros(3, np.float32)
        p, n = np.asarray(p, np.float32), np.asarray(n, np.float32)
        self.lib.orc_offset_ray_origin(_ptr(p), _ptr(n), _ptr(out))
        return out

    def power_heuristic(self, f: float, g: float) -> float:
        return float(self.lib.orc_power_heuristic(f, g))

    def triangle(self, p9):
        out = np.zeros(12, np.float32)
        area = np.zeros(1, np.float32)
        p9 = np.asarray(p9, np.float32)
        self.lib.orc_triangle(_ptr(p9), _ptr(out), _ptr(area))
        return out, float(area[0])

    def triangle_intersect(self, p9, o, d, tmax=3.4028234663852886e38):
        tuv = np.zeros(3, np.float32)
        p9, o, d = (np.asarray(v, np.float32) for v in (p9, o, d))
        hit = self.lib.orc_triangle_intersect(_ptr(p9), _ptr(o), _ptr(d), tmax, _ptr(tuv))
        return bool(hit), tuv

    def sincos(self, x: float):
        s = np.zeros(1, np.float32)
        c = np.zeros(1, np.float32)
        self.lib.orc_sincos(x, _ptr(s), _ptr(c))
        return float(s[0]), float(c[0])

    def pow5(self, x: float) -> float:
        return float(self.lib.orc_pow5(x))

    def sample_f(self, material, wo, n, state):
        out = np.zeros(10, np.float32)
        material = np.ascontiguousarray(material)
        wo, n = np.asarray(wo, np.float32), np.asarray(n, np.float32)
        self.lib.orc_sample_f(_ptr(material), _ptr(wo), _ptr(n), _ptr(state), _ptr(out))
        return out

    # ------------------------------------------------------------------ ray log (traversal audit)
    def raylog_enable(self, on: bool = True) -> None:
        """Log every shadow ray and path ray the following renders trace (cleared by each call)."""
        self.lib.orc_raylog_enable(int(on))

    def raylog_fetch(self) -> dict:
        cnt = np.zeros(2, np.int64)
        self.lib.orc_raylog_counts(_ptr(cnt))
        na, nc = int(cnt[0]), int(cnt[1])
        odt = np.zeros((max(na, 1), 7), np.float32)
        info = np.zeros((max(na, 1), 2), np.int32)
        od = np.zeros((max(nc, 1), 6), np.float32)
        tri = np.zeros(max(nc, 1), np.int32)
        t = np.zeros(max(nc, 1), np.float32)
        self.lib.orc_raylog_fetch_any(_ptr(odt), _ptr(info))
        self.lib.orc_raylog_fetch_closest(_ptr(od), _ptr(tri), _ptr(t))
        return {"any_o": odt[:na, 0:3].copy(), "any_d": odt[:na, 3:6].copy(), "any_tmax": odt[:na, 6].copy(),
                "any_excluded": info[:na, 0].copy(), "any_occluded": info[:na, 1].copy(),
                "closest_o": od[:nc, 0:3].copy(), "closest_d": od[:nc, 3:6].copy(), "closest_tri": tri[:nc].copy(),
                "closest_t": t[:nc].copy()}

    # ------------------------------------------------------------------ scene
    def scene(self, arrays) -> "OracleScene":
        return OracleScene(self, arrays)


class OracleScene:
    def __init__(self, oracle: Oracle, arrays):
        self.o = oracle
        self.arrays = arrays
        L = oracle.lib
        tris = np.ascontiguousarray(arrays.tris, np.float32)
        tm = np.ascontiguousarray(arrays.tri_material, np.int32)
        tl = np.ascontiguousarray(arrays.tri_light, np.int32)
        mats = np.ascontiguousarray(arrays.materials)
        lights = np.ascontiguousarray(arrays.lights)
        self.h = L.orc_scene_create(_ptr(tris), tris.shape[0], _ptr(tm), _ptr(tl), _ptr(mats), mats.shape[0],
                                    _ptr(lights), lights.shape[0])

    def set_watertight(self, on: bool = True) -> "OracleScene":
        """Conservative box decisions instead of the reference's fp32 slab test (oracle.cpp, AabbIsect): the BVH walk
        then returns what exhaustive search returns.  Default off = the literal reference."""
        self.o.lib.orc_scene_set_watertight(self.h, int(on))
        return self

    def close(self):
        if self.h:
            self.o.lib.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh_stats(self) -> dict:
        out = np.zeros(12, np.int64)
        root = np.zeros(6, np.float32)
        self.o.lib.orc_scene_bvh_stats(self.h, _ptr(out), _ptr(root))
        return {"num_nodes": int(out[0]), "num_prims": int(out[1]), "max_depth": int(out[2]),
                "num_leaves": int(out[3]), "leaf_hist": [int(x) for x in out[4:12]], "root_bounds": root}

    def nodes(self):
        """The restated reference BVH (bvh.cuh:30-219): (bounds (n, 6), num_primitives (n,), index (n,), primitive order)."""
        st = self.bvh_stats()
        n, m = st["num_nodes"], st["num_prims"]
        bounds = np.zeros((n, 6), np.float32)
        count, index, prim_tri = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(max(m, 1), np.int32)
        self.o.lib.orc_scene_nodes(self.h, _ptr(bounds), _ptr(count), _ptr(index), _ptr(prim_tri))
        return bounds, count, index, prim_tri[:m]

    def trace_closest(self, o3, d3, tmax, threads: int = 8):
        o3 = np.ascontiguousarray(o3, np.float32)
        d3 = np.ascontiguousarray(d3, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = o3.shape[0]
        tri = np.zeros(n, np.int32)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        self.o.lib.orc_trace_closest(self.h, n, _ptr(o3), _ptr(d3), _ptr(tmax), _ptr(tri), _ptr(t), _ptr(u),
                                     _ptr(v), threads)
        return tri, t, u, v

    def trace_closest_brute(self, o3, d3, tmax, threads: int = 8):
        o3 = np.ascontiguousarray(o3, np.float32)
        d3 = np.ascontiguousarray(d3, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = o3.shape[0]
        tri = np.zeros(n, np.int32)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        self.o.lib.orc_trace_closest_brute(self.h, n, _ptr(o3), _ptr(d3), _ptr(tmax), _ptr(tri), _ptr(t),
                                           _ptr(u), _ptr(v), threads)
        return tri, t, u, v

    def closest_ties(self, o, d, tmax, t_ref):
        o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
        tris = np.zeros(16, np.int32)
        k = self.o.lib.orc_closest_ties(self.h, _ptr(o), _ptr(d), tmax, t_ref, _ptr(tris), 16)
        return [int(x) for x in tris[:min(k, 16)]]

    def trace_any(self, o3, d3, tmax, excluded, threads: int = 8):
        o3 = np.ascontiguousarray(o3, np.float32)
        d3 = np.ascontiguousarray(d3, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        excluded = np.ascontiguousarray(excluded, np.int32)
        n = o3.shape[0]
        occ = np.zeros(n, np.int32)
        self.o.lib.orc_trace_any(self.h, n, _ptr(o3), _ptr(d3), _ptr(tmax), _ptr(excluded), _ptr(occ), threads)
        return occ

    def trace_any_brute(self, o3, d3, tmax, excluded, threads: int = 8):
        """Exhaustive any hit: index of the first accepted, non-excluded triangle (original order) or -1."""
        o3 = np.ascontiguousarray(o3, np.float32)
        d3 = np.ascontiguousarray(d3, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        excluded = np.ascontiguousarray(excluded, np.int32)
        n = o3.shape[0]
        occ = np.zeros(n, np.int32)
        self.o.lib.orc_trace_any_brute(self.h, n, _ptr(o3), _ptr(d3), _ptr(tmax), _ptr(excluded), _ptr(occ), threads)
        return occ

    def render(self, cam12, width, height, spp, max_bounces=10, seed=1, slot_lo=0, slot_hi=1 << 20,
               threads=1, collect_stats=False, iter_cap=4096, fixed_out=None):
        """Literal wavefront render.  Returns (image (h,w,3) post-processed, raw sums (h,w,3), stats).
        ``fixed_out``: a zeroed (h, w, 3) int64 array that also receives the product's fixed-point accumulation of the frame
        (RT_FLAG_DETERMINISTIC's sums: oracle.cpp render_literal)."""
        cam12 = np.ascontiguousarray(cam12, np.float32)
        fb_sum = np.zeros((height, width, 3), np.float32)
        fb_out = np.zeros((height, width, 3), np.float32)
        st = np.zeros(20, np.float64)
        it = np.zeros((iter_cap, 4), np.int32)
        if fixed_out is not None:
            assert fixed_out.dtype == np.int64 and fixed_out.shape == (height, width, 3) and fixed_out.flags["C_CONTIGUOUS"]
            self.o.lib.orc_render_also_fixed.argtypes = [ctypes.c_void_p]
            self.o.lib.orc_render_also_fixed(_ptr(fixed_out))
        self.o.lib.orc_render(self.h, _ptr(cam12), width, height, spp, max_bounces, seed, slot_lo, slot_hi,
                              threads, int(collect_stats), _ptr(fb_sum), _ptr(fb_out), _ptr(st), _ptr(it),
                              iter_cap)
        names = ["iterations", "sum_mat", "sum_gen", "sum_ah", "sum_ch", "emission_adds", "ah_adds", "ch_adds",
                 "rr_draws", "rr_kills", "seconds_loop", "seconds_rng_init", "ch_rays", "ch_node_pairs",
                 "ch_tri_tests", "ah_rays", "ah_node_pairs", "ah_tri_tests", "max_stack", "n_iter_records"]
        stats = {k: (float(v) if k.startswith("seconds") else int(v)) for k, v in zip(names, st)}
        stats["iter_counts"] = it[: stats["n_iter_records"]].copy()
        return fb_out, fb_sum, stats


def fnv1a64_bytes(a: np.ndarray) -> int:
    h = 0xCBF29CE484222325
    for b in a.tobytes():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def fnv1a64_words(a: np.ndarray) -> int:
    """FNV-1a-64 folding one 32-bit word per step."""
    h = 0xCBF29CE484222325
    for w in a.view(np.uint32).ravel().tolist():
        h ^= w
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def sums_hash(fixed: np.ndarray) -> str:
    """64-bit hash of a frame's int64 fixed-point sums (the first 16 hex digits of SHA-256 over the little-endian bytes, row-major,
    top row first, RGB interleaved): what tests/golden/full_size_image_hashes.json holds per frame and oracle mode."""
    import hashlib
    a = np.ascontiguousarray(fixed, dtype="<i8")
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]
