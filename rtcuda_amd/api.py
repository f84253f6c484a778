"""ctypes binding of ``librtcuda_amd.so`` (the C-ABI in ``include/rtcuda_amd.h``).

Mirrors the reference's host interface for the render path -- ``Scene`` construction from
triangles / materials / lights, ``Camera(lookfrom, lookat, up, vfov, aspect)`` and
``render(width, height, spp, max_bounces, camera, scene)`` (render.cuh:366-367) -- on top of the
library.  Nothing here computes pixels: every call goes to the HIP library and raises
``RtError`` if it fails or ``ImportError`` if the library has not been built.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from .scenes import SceneArrays

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, os.environ.get("RT_LIB_NAME", "librtcuda_amd.so"))  # RT_LIB_NAME: instrumented dev builds
CSRC = os.path.join(_PKG, "csrc")

W = 1 << 20  # RT_NUM_WORKING_PATHS
FLAG_TIME_KERNELS = 1
FLAG_DETERMINISTIC = 2
FLAG_RNG_PER_SAMPLE = 4  # NOT the reference's random numbers (see include/rtcuda_amd.h): partition-invariant streams
FLAG_REFERENCE_WALK = 8  # cross-check mode: every ray walks the reference's own tree literally (slow; the default kernels give the same image)
FLAG_WATERTIGHT = 16     # the triangle-list definition (no hit lost to a box test, ties by caller index) instead of the reference's

EXPORTS = [
    "rt_scene_create", "rt_scene_destroy", "rt_scene_info", "rt_scene_build_info", "rt_camera_make", "rt_render", "rt_render_multi",
    "rt_render_shard", "rt_render_shard_fixed", "rt_post_process", "rt_post_process_fixed", "rt_trace_closest", "rt_trace_any",
    "rt_trace_closest_flags", "rt_trace_any_flags", "rt_xorwow_states", "rt_shutdown", "rt_peer_access_log", "rt_last_error", "rt_version", "rt_build_id",
]
# the lab (include/rtcuda_amd_tools.h, librtcuda_amd_tools.so): measurement tools, not part of the drop-in C-ABI
TOOLS_LIB_PATH = os.path.join(_PKG, "librtcuda_amd_tools.so")
TOOLS_EXPORTS = ["rt_measure_copy_bandwidth", "rt_calibrate_valu", "rt_calibrate_valu_packed", "rt_probe_issue", "rt_split_probe"]


class RtError(RuntimeError):
    pass


class RtStats(ctypes.Structure):
    _fields_ = [
        ("camera_rays", ctypes.c_int64), ("shade_events", ctypes.c_int64), ("closest_rays", ctypes.c_int64),
        ("any_rays", ctypes.c_int64), ("emission_adds", ctypes.c_int64), ("shadow_adds", ctypes.c_int64),
        ("rr_draws", ctypes.c_int64), ("iterations", ctypes.c_int64), ("bvh_nodes", ctypes.c_int64),
        ("bvh_depth", ctypes.c_int64), ("seconds_render", ctypes.c_double), ("seconds_rng_init", ctypes.c_double),
        ("seconds_trace", ctypes.c_double), ("seconds_reference_tree", ctypes.c_double), ("seconds_advance", ctypes.c_double),
        ("launches_trace", ctypes.c_int64), ("reserved", ctypes.c_int64 * 7),
    ]

    def as_dict(self) -> dict:
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}
        d["lds_top_records"] = int(self.reserved[2])  # BVH records the persistent kernel staged in LDS
        # default kernels (no RT_FLAG_WATERTIGHT): closest hits re-traced through the reference's own tree, accepted hits the
        # reference's box test loses, exact ties at the final distance
        d["literal_retraces"], d["reference_lost_hits"], d["exact_ties"] = (int(self.reserved[k]) for k in (4, 5, 6))
        return d


def build(verbose: bool = False) -> str:
    """Compile the library in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC], stdout=out)
    return LIB_PATH


_lib = None


def _preload_hip_runtime() -> None:
    """Make ONE HIP runtime globally visible before the library is loaded.

    librtcuda_amd.so deliberately carries no DT_NEEDED on libamdhip64 (see csrc/Makefile): a
    PyTorch-ROCm wheel ships its own libamdhip64.so, and a second runtime in the same process
    finds no GPU.  When torch is installed its copy is used (bench.py needs torch.distributed in
    the same process); otherwise the system ROCm runtime.
    """
    candidates = []
    try:
        import torch  # noqa: F401
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except ImportError:
        pass
    candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    for c in candidates:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            return
        except OSError:
            continue
    raise ImportError("no HIP runtime (libamdhip64.so) could be loaded")


def lib():
    """Load the library (once).  Raises ImportError with the build command if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is the product and there is no fallback. "
            f"Build it with `make -C {CSRC}` (or python -c 'import __graft_entry__ as g; g.build()').")
    _preload_hip_runtime()
    _lib = _bind(ctypes.CDLL(LIB_PATH))
    return _lib


def _bind(L):
    """Argument types of the drop-in C-ABI (include/rtcuda_amd.h) on a loaded library."""
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    L.rt_last_error.restype = ctypes.c_char_p
    L.rt_version.restype = ctypes.c_char_p
    L.rt_build_id.restype = ctypes.c_char_p
    L.rt_scene_create.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, ctypes.POINTER(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_destroy.restype = None
    L.rt_scene_info.argtypes = [vp, vp]
    L.rt_scene_build_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_double)]
    L.rt_camera_make.argtypes = [vp, vp, vp, cf, cf, vp]
    L.rt_render.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_uint64, ctypes.c_uint32, vp, ctypes.POINTER(RtStats)]
    L.rt_render_multi.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_uint64, ctypes.c_uint32, vp, ci, vp, ctypes.POINTER(RtStats)]
    L.rt_render_shard.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_uint64, ci, ci, ctypes.c_uint32, vp, vp,
                                  ctypes.POINTER(RtStats)]
    L.rt_post_process.argtypes = [vp, ci, ci, vp]
    L.rt_render_shard_fixed.argtypes = L.rt_render_shard.argtypes
    L.rt_post_process_fixed.argtypes = [vp, vp, ci, ci, vp]
    L.rt_trace_closest.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    L.rt_trace_any.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    L.rt_trace_closest_flags.argtypes = [vp, ctypes.c_uint32, ci, vp, vp, vp, vp, vp, vp, vp]
    L.rt_trace_any_flags.argtypes = [vp, ctypes.c_uint32, ci, vp, vp, vp, vp, vp]
    L.rt_xorwow_states.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ci, vp, vp]
    L.rt_shutdown.restype = None
    L.rt_peer_access_log.restype = ctypes.c_char_p
    return L


_tools = None


def tools_lib():
    """The LAB library (librtcuda_amd_tools.so: the product's translation unit + the measurement tools of
    include/rtcuda_amd_tools.h).  Loaded on first use by bench.py's roofline block and tools/*.py -- never by a render."""
    global _tools
    if _tools is not None:
        return _tools
    if not os.path.exists(TOOLS_LIB_PATH):
        raise ImportError(f"{TOOLS_LIB_PATH} not found: build it with `make -C {CSRC}`")
    _preload_hip_runtime()
    L = _bind(ctypes.CDLL(TOOLS_LIB_PATH))
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.rt_measure_copy_bandwidth.argtypes = [ctypes.c_int64, ci, ctypes.POINTER(ctypes.c_double)]
    L.rt_calibrate_valu.argtypes = [ci, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.rt_calibrate_valu_packed.argtypes = [ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
    L.rt_probe_issue.argtypes = [ci, ci, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.rt_split_probe.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_uint64, ctypes.c_int64, vp, ci]
    _tools = L
    return L


def shutdown() -> None:
    """Release the library's hidden device allocations (render contexts, cached output buffers): rt_shutdown."""
    lib().rt_shutdown()


def peer_access_log() -> str:
    """rt_render_multi: what became of peer access between the listed devices (rt_peer_access_log)."""
    return lib().rt_peer_access_log().decode()


def build_id() -> str:
    """Hash of the sources + flags the LOADED library's device code was built from (rt_build_id)."""
    return lib().rt_build_id().decode()


def _check(rc: int, what: str, library=None) -> None:
    if rc != 0:
        raise RtError(f"{what}: {(library or lib()).rt_last_error().decode(errors='replace')}")


def _p(a):
    return None if a is None else a.ctypes.data


def make_camera(lookfrom=(0.5, 0.5, 1.5), lookat=(0.5, 0.5, 0.0), up=(0.0, 1.0, 0.0), vfov=37.8,
                aspect=1.0) -> np.ndarray:
    """``Camera(lookfrom, lookat, up, vfov, aspect)`` (camera.cuh:15-29) -> the 12-float POD."""
    a, b, c = (np.asarray(v, np.float32) for v in (lookfrom, lookat, up))
    out = np.zeros(12, np.float32)
    _check(lib().rt_camera_make(_p(a), _p(b), _p(c), float(vfov), float(aspect), _p(out)), "rt_camera_make")
    return out


class Scene:
    """Device-resident scene (triangles, materials, lights, BVH) on the current HIP device."""

    def __init__(self, arrays: SceneArrays, library=None):
        """`library`: the loaded library that owns the scene -- the product (default) or `tools_lib()`, whose private copy of
        the product's entry points is what rt_split_probe works on."""
        L = self.L = library or lib()
        self.arrays = arrays
        tris = np.ascontiguousarray(arrays.tris, np.float32)
        tm = np.ascontiguousarray(arrays.tri_material, np.int32)
        tl = np.ascontiguousarray(arrays.tri_light, np.int32)
        mats = np.ascontiguousarray(arrays.materials)
        lights = np.ascontiguousarray(arrays.lights)
        assert mats.dtype.itemsize == 20 and lights.dtype.itemsize == 32
        h = ctypes.c_void_p()
        _check(L.rt_scene_create(_p(tris), tris.shape[0], _p(tm), _p(tl), _p(mats), mats.shape[0], _p(lights),
                                 lights.shape[0], ctypes.byref(h)), "rt_scene_create", L)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        out = np.zeros(4, np.int64)
        _check(self.L.rt_scene_info(self.h, _p(out)), "rt_scene_info", self.L)
        b, sec = ctypes.c_int(0), ctypes.c_double(0.0)
        _check(self.L.rt_scene_build_info(self.h, ctypes.byref(b), ctypes.byref(sec)), "rt_scene_build_info", self.L)
        return {"pairs": int(out[0]), "tris": int(out[1]), "max_depth": int(out[2]), "leaves": int(out[3]),
                "builder": "lbvh" if b.value else "sah", "build_seconds": sec.value}

    # ---- render(): the drop-in entry point
    def render(self, camera: np.ndarray, width: int, height: int, spp: int, max_bounces: int = 10,
               seed: int = 1, flags: int = 0):
        """Whole frame on the current device -> (image (h, w, 3) float32 post-processed, stats)."""
        cam = np.ascontiguousarray(camera, np.float32)
        out = np.zeros((height, width, 3), np.float32)
        st = RtStats()
        _check(self.L.rt_render(self.h, _p(cam), width, height, spp, max_bounces, seed, flags, _p(out),
                               ctypes.byref(st)), "rt_render", self.L)
        return out, st.as_dict()

    def render_multi(self, camera: np.ndarray, width: int, height: int, spp: int, devices, max_bounces: int = 10,
                     seed: int = 1, flags: int = 0):
        """Whole frame over the listed devices in this one process (rt_render_multi) -> (image, stats)."""
        cam = np.ascontiguousarray(camera, np.float32)
        dev = np.ascontiguousarray(devices, np.int32)
        out = np.zeros((height, width, 3), np.float32)
        st = RtStats()
        _check(self.L.rt_render_multi(self.h, _p(cam), width, height, spp, max_bounces, seed, flags, _p(dev), int(dev.shape[0]),
                                     _p(out), ctypes.byref(st)), "rt_render_multi", self.L)
        d = st.as_dict()
        d["device_shards"] = int(st.reserved[3])
        return out, d

    def render_shard(self, camera: np.ndarray, width: int, height: int, spp: int, shard_index: int,
                     shard_count: int, d_sum_ptr: int, max_bounces: int = 10, seed: int = 1, flags: int = 0,
                     stream: int = 0) -> dict:
        """Adds this shard's raw sums into the DEVICE buffer at ``d_sum_ptr`` (w*h*3 floats)."""
        cam = np.ascontiguousarray(camera, np.float32)
        st = RtStats()
        _check(self.L.rt_render_shard(self.h, _p(cam), width, height, spp, max_bounces, seed, shard_index,
                                     shard_count, flags, ctypes.c_void_p(d_sum_ptr), ctypes.c_void_p(stream),
                                     ctypes.byref(st)), "rt_render_shard", self.L)
        return st.as_dict()

    def render_shard_fixed(self, camera: np.ndarray, width: int, height: int, spp: int, shard_index: int,
                           shard_count: int, d_sum_fixed_ptr: int, max_bounces: int = 10, seed: int = 1, flags: int = 0,
                           stream: int = 0) -> dict:
        """Order-independent accumulation: adds int64 fixed-point sums (2^-30) into the DEVICE buffer."""
        cam = np.ascontiguousarray(camera, np.float32)
        st = RtStats()
        _check(self.L.rt_render_shard_fixed(self.h, _p(cam), width, height, spp, max_bounces, seed, shard_index,
                                           shard_count, flags, ctypes.c_void_p(d_sum_fixed_ptr), ctypes.c_void_p(stream),
                                           ctypes.byref(st)), "rt_render_shard_fixed", self.L)
        return st.as_dict()

    # ---- stage-level entry points (parity tests)
    def trace_closest(self, o3, d3, tmax, flags: int = 0):
        """``flags=FLAG_REFERENCE_WALK``: through the reference's own tree and walk (rt_trace_closest_flags)."""
        o3 = np.ascontiguousarray(o3, np.float32)
        d3 = np.ascontiguousarray(d3, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = o3.shape[0]
        tri = np.zeros(n, np.int32)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        if flags:
            _check(self.L.rt_trace_closest_flags(self.h, flags, n, _p(o3), _p(d3), _p(tmax), _p(tri), _p(t), _p(u), _p(v)),
                   "rt_trace_closest_flags", self.L)
        else:
            _check(self.L.rt_trace_closest(self.h, n, _p(o3), _p(d3), _p(tmax), _p(tri), _p(t), _p(u), _p(v)),
                   "rt_trace_closest", self.L)
        return tri, t, u, v

    def trace_any(self, o3, d3, tmax, excluded, flags: int = 0):
        o3 = np.ascontiguousarray(o3, np.float32)
        d3 = np.ascontiguousarray(d3, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        excluded = np.ascontiguousarray(excluded, np.int32)
        n = o3.shape[0]
        occ = np.zeros(n, np.int32)
        if flags:
            _check(self.L.rt_trace_any_flags(self.h, flags, n, _p(o3), _p(d3), _p(tmax), _p(excluded), _p(occ)),
                   "rt_trace_any_flags", self.L)
        else:
            _check(self.L.rt_trace_any(self.h, n, _p(o3), _p(d3), _p(tmax), _p(excluded), _p(occ)), "rt_trace_any", self.L)
        return occ


def post_process(d_ptr: int, num_pixels: int, spp: int, stream: int = 0) -> None:
    _check(lib().rt_post_process(ctypes.c_void_p(d_ptr), num_pixels, spp, ctypes.c_void_p(stream)),
           "rt_post_process")


def post_process_fixed(d_fixed_ptr: int, d_out_ptr: int, num_pixels: int, spp: int, stream: int = 0) -> None:
    _check(lib().rt_post_process_fixed(ctypes.c_void_p(d_fixed_ptr), ctypes.c_void_p(d_out_ptr), num_pixels, spp,
                                       ctypes.c_void_p(stream)), "rt_post_process_fixed")


def xorwow_states(seed: int, first: int, count: int, draws: int = 0):
    st = np.zeros((count, 6), np.uint32)
    uni = np.zeros((count, max(draws, 1)), np.float32)
    _check(lib().rt_xorwow_states(seed, first, count, draws, _p(st), _p(uni)), "rt_xorwow_states")
    return st, uni[:, :draws]


def measure_copy_bandwidth(nbytes: int = 1 << 30, reps: int = 5) -> float:
    out = ctypes.c_double(0.0)
    _check(tools_lib().rt_measure_copy_bandwidth(nbytes, reps, ctypes.byref(out)), "rt_measure_copy_bandwidth", tools_lib())
    return out.value


def calibrate_valu(waves_per_simd: int = 4, iters: int = 20000):
    """(lane-operations/s the vector ALUs sustain on independent v_fma_f32, wave-instructions per launch)."""
    rate, winstr = ctypes.c_double(0.0), ctypes.c_double(0.0)
    _check(tools_lib().rt_calibrate_valu(waves_per_simd, iters, ctypes.byref(rate), ctypes.byref(winstr)), "rt_calibrate_valu", tools_lib())
    return rate.value, winstr.value


def calibrate_valu_packed(waves_per_simd: int = 4, iters: int = 20000, kind: int = 1) -> float:
    """Lane-operations/s of a packed-fp32 stream (kind 1 v_pk_fma_f32, 2 v_pk_mul_f32, 3 v_pk_add_f32; 2 per lane and instruction)."""
    rate = ctypes.c_double(0.0)
    _check(tools_lib().rt_calibrate_valu_packed(waves_per_simd, iters, kind, ctypes.byref(rate)), "rt_calibrate_valu_packed", tools_lib())
    return rate.value


PROBE_ISSUE_KINDS = ["v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_xor_b32", "v_lshlrev_b32", "v_max_f32",
                     "v_rcp_f32", "v_sqrt_f32", "v_cndmask_b32", "v_mul_f32 -> v_add_f32 (dependent pair)", "v_fma_f32, one dependent chain",
                     "v_mul_f32, one dependent chain", "v_mul_f32 literal", "v_mul_f32 sgpr", "v_fma_f32 2 sgprs",
                     "v_cndmask_b32 sgpr-pair mask", "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 -> v_cndmask_b32 (dependent pair)", "v_bfi_b32",
                     "v_and_b32", "mix 3 v_mul + 1 v_cndmask", "mix 3 v_mul + 1 v_max", "mix 4 v_mul", "mix 3 v_mul + 1 v_mov"]


def probe_issue(kind: int, waves_per_simd: int, iters: int = 20000):
    """(seconds of the best launch, instructions every wave issued) of the issue probe (rt_probe_issue)."""
    sec, n = ctypes.c_double(0.0), ctypes.c_double(0.0)
    _check(tools_lib().rt_probe_issue(kind, waves_per_simd, iters, ctypes.byref(sec), ctypes.byref(n)), "rt_probe_issue", tools_lib())
    return sec.value, n.value


# rt_split_probe's out[] layout (the RT_PROBE_* enum of include/rtcuda_amd.h)
PROBE_FIELDS = (["rounds", "closest_rays", "any_rays", "s_advance_round0", "s_advance", "s_trace_pool"]
                + [f"s_trace_closest_w{w}" for w in (8, 6, 5, 4)] + [f"s_trace_any_w{w}" for w in (8, 6, 5, 4)]
                + [f"trace_blocks_per_cu_w{w}" for w in (8, 6, 5, 4)]
                + [f"shades_{k}" for k in ("matte", "mirror", "glass")] + [f"s_shade_{k}" for k in ("matte", "mirror", "glass")])


def split_probe(scene: "Scene", camera: np.ndarray, width: int, height: int, spp: int, target_rays: int,
                max_bounces: int = 10, seed: int = 1) -> dict:
    """Trace-only and shade-only rates on rays / shading records dumped from the round pipeline (rt_split_probe)."""
    cam = np.ascontiguousarray(camera, np.float32)
    out = np.zeros(len(PROBE_FIELDS), np.float64)
    if scene.L is not tools_lib():
        raise RtError("split_probe: the scene must be created with library=tools_lib()")
    _check(tools_lib().rt_split_probe(scene.h, _p(cam), width, height, spp, max_bounces, seed, target_rays, _p(out), len(out)),
           "rt_split_probe", tools_lib())
    return dict(zip(PROBE_FIELDS, out.tolist()))


def render(width: int, height: int, num_samples: int, max_bounces: int, camera: np.ndarray, scene: Scene,
           seed: int = 1):
    """Same argument order as the reference's ``render()`` (render.cuh:366-367); returns the framebuffer."""
    img, _ = scene.render(camera, width, height, num_samples, max_bounces, seed)
    return img
