"""CPU tests of the host side: C-ABI surface, scene recipes, PLY ingest, BVH builder, sharding."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

import raygen
from conftest import ROOT, default_camera
from rtcuda_amd import dist as rtdist
from rtcuda_amd import scenes

with open(os.path.join(ROOT, "tests", "golden", "appendix_c.json")) as _fh:
    import json
    APPX = json.load(_fh)


# ----------------------------------------------------------------------------- C-ABI surface
def _declared_functions(header="rtcuda_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads on a machine without a GPU and exports exactly what the header declares."""
    from rtcuda_amd import api
    api.build()
    declared = _declared_functions()
    assert set(declared) == set(api.EXPORTS), (declared, api.EXPORTS)
    L = api.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.rt_version()
    # the drop-in library carries the drop-in C-ABI and nothing else: the measurement tools live in a library of their own
    tools = _declared_functions("rtcuda_amd_tools.h")
    assert set(tools) == set(api.TOOLS_EXPORTS), (tools, api.TOOLS_EXPORTS)
    T = api.tools_lib()
    for name in tools:
        assert hasattr(T, name), name
        assert not hasattr(L, name), f"{name} is a lab entry point and must not be exported by the product library"


def test_struct_layouts_match_the_header():
    from rtcuda_amd import api
    assert scenes.MATERIAL_DTYPE.itemsize == 20   # rt_material / material.cuh:20-22
    assert scenes.LIGHT_DTYPE.itemsize == 32      # rt_light
    assert ctypes.sizeof(api.RtStats) == 8 * 10 + 8 * 5 + 8 + 8 * 7


def test_camera_make_needs_no_gpu(oracle):
    from rtcuda_amd import api
    for aspect in (1.0, 16 / 9, 0.75):
        assert np.array_equal(api.make_camera(aspect=aspect).view(np.uint32),
                              default_camera(oracle, aspect).view(np.uint32))


def test_compute_calls_fail_loudly_without_a_gpu(bunny_matte):
    """No CPU fallback: without a device the product raises instead of computing somewhere else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rtcuda_amd import api
    with pytest.raises(api.RtError):
        api.Scene(bunny_matte)


# ----------------------------------------------------------------------------- scene recipes
def test_bunny_transform_matches_known_rows():
    m = scenes.bunny_transform()
    for got, ref in zip(m[:3], APPX["bunny_transform_rows"]):
        assert np.allclose(got, ref, rtol=0, atol=5e-9)
    assert np.array_equal(m[3], np.array([0, 0, 0, 1], np.float32))


def test_cornell_bunny_arrays(bunny_matte):
    a = bunny_matte
    assert a.n_tris == 69463 and a.tris.dtype == np.float32 and a.tris.shape == (69463, 9)
    assert (a.tri_material[:69451] == 3).all()                      # brown bunny (main.cu:83)
    assert a.tri_material[69451:69461].tolist() == [0, 0, 1, 1, 2, 2, 2, 2, 2, 2]
    # light order of the reference's unordered_map iteration: [69462, 69461]
    assert a.lights["tri"].tolist() == [69462, 69461]
    assert a.tri_light[69462] == 0 and a.tri_light[69461] == 1 and (a.tri_light[:69461] == -1).all()
    assert np.allclose(a.lights["L"], 15.0)
    lo, hi = a.tris.reshape(-1, 3).min(0), a.tris.reshape(-1, 3).max(0)
    assert np.array_equal(lo, [0, 0, -1]) and np.array_equal(hi, [1, 1, 0])  # "Global bounding box"
    bun = a.tris[:69451].reshape(-1, 3)
    assert np.allclose(bun.min(0), [0.300, 0.0, -0.741], atol=2e-3) and np.allclose(bun.max(0), [0.611, 0.309, -0.500], atol=2e-3)


def test_scene_variants():
    full = scenes.cornell_bunny("full_bsdf")
    assert (full.tri_material[:69451] == 4).all() and full.materials[4]["type"] == scenes.GLASS
    assert full.tri_material[69459:69461].tolist() == [5, 5] and full.materials[5]["type"] == scenes.MIRROR
    four = scenes.cornell_bunny("four_bunnies")
    assert four.n_tris == 4 * 69451 + 12 == 277816
    sixteen = scenes.cornell_bunny("sixteen_lights")
    assert sixteen.n_tris == 69477 and len(sixteen.lights) == 16
    assert sixteen.lights["tri"].tolist() == list(range(69461, 69477))  # ascending triangle index
    box = scenes.cornell_bunny("matte", bunny=False)
    assert box.n_tris == 12 and box.lights["tri"].tolist() == [11, 10]
    with pytest.raises(ValueError):
        scenes.cornell_bunny("nope")


def test_ply_reader_ascii_and_binary(tmp_path):
    pos, faces = scenes.load_ply()
    assert pos.shape == (35947, 3) and faces.shape == (69451, 3) and faces.max() == 35946
    assert np.array_equal(pos[0], np.array([-0.0378297, 0.12794, 0.00447467], np.float32))
    # the same mesh as binary_little_endian with a different property order and an extra element
    path = tmp_path / "tiny.ply"
    v = pos[:5]
    f = np.array([[0, 1, 2], [2, 3, 4]], np.int32)
    with open(path, "wb") as fh:
        fh.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float z\nproperty float x\n"
                 b"property float y\nproperty uchar flag\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n")
        for p in v:
            fh.write(struct.pack("<fffB", p[2], p[0], p[1], 7))
        for t in f:
            fh.write(struct.pack("<Biii", 3, *t))
    p2, f2 = scenes.load_ply(str(path))
    assert np.array_equal(p2, v) and np.array_equal(f2, f)
    bad = tmp_path / "quad.ply"
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                   "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    with pytest.raises(ValueError):
        scenes.load_ply(str(bad))


def test_ppm_quantisation(tmp_path):
    img = np.array([[[0.0, 0.5, 1.0], [2.0, -1.0, 0.999]]], np.float32)
    path = tmp_path / "o.ppm"
    scenes.write_ppm(str(path), img)
    lines = path.read_text().split("\n")
    assert lines[:3] == ["P3", "2 1", "255"]
    assert lines[3] == "0 128 255" and lines[4] == "255 0 255"   # clamp(int(256 c), 0, 255), main.cu:186-188


# ----------------------------------------------------------------------------- BVH builder (host, no GPU)
def _hostcheck():
    from rtcuda_amd import api
    api.build()
    L = ctypes.CDLL(os.path.join(os.path.dirname(api.LIB_PATH), "librt_hostcheck.so"))
    L.rt_bvh_selfcheck.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p]
    return L


def _selfcheck(L, tris, ro, rd):
    out = np.zeros(10, np.int64)
    tris = np.ascontiguousarray(tris, np.float32)
    ro = np.ascontiguousarray(ro, np.float32)
    rd = np.ascontiguousarray(rd, np.float32)
    L.rt_bvh_selfcheck(tris.ctypes.data, tris.shape[0], ro.shape[0], ro.ctypes.data if len(ro) else None,
                       rd.ctypes.data if len(rd) else None, out.ctypes.data)
    return dict(zip(["nodes", "leaves", "depth", "maxleaf", "errors", "mismatch", "maxstack", "maxsteps", "stack_bound",
                     "bin_depth"], out.tolist()))


def test_product_bvh_structure_and_cpu_walk(oracle, bunny_matte):
    L = _hostcheck()
    cam = default_camera(oracle, 16 / 9)
    ro, rd = raygen.camera_rays(cam, 1920, 1080, 600, seed=3)
    ao, ad = raygen.axis_aligned_rays(300, seed=4)
    r = _selfcheck(L, bunny_matte.tris, np.concatenate([ro, ao]), np.concatenate([rd, ad]))
    assert r["errors"] == 0 and r["mismatch"] == 0
    assert r["depth"] <= 16 and r["maxleaf"] <= 4 and r["maxstack"] <= r["stack_bound"] <= 160
    assert r["nodes"] < r["leaves"]  # 4-wide: fewer node records than leaves


@pytest.mark.parametrize("scale", [3.0, 30.0])
def test_product_walk_finds_every_hit_of_rays_that_start_far_outside_the_scene(bunny_matte, scale):
    """The 4-wide node step computes a plane distance as ONE fma, b * (1 / d) - o * (1 / d); the rounding of o * (1 / d) moves
    the planes of an axis by up to 2^-24 |o|, so the records are padded for the origins that are traced (rt_bvh.h,
    pad_quads_for_origins; the selfcheck validates the padded records and their margins).  Rays from up to ~100 scene sizes
    away, aimed at points near the vertices and edges of random triangles: both record formats find what exhaustive search
    finds.  (From thousands of scene sizes away the fp32 triangle test itself is noisier than any box: not a regime the
    reference's scenes have.)"""
    L = _hostcheck()
    tris = np.asarray(bunny_matte.tris, np.float32).reshape(-1, 9)[::16]
    rng = np.random.default_rng(5)
    n = 20000
    tsel = tris[rng.integers(0, len(tris), n)].reshape(n, 3, 3)
    w = rng.dirichlet([0.3, 0.3, 0.3], n).astype(np.float32)
    target = (tsel * w[:, :, None]).sum(axis=1)
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    ro = (target - dirs * scale * rng.uniform(0.3, 3.0, (n, 1))).astype(np.float32)
    rd = target - ro.astype(np.float64)
    rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    r = _selfcheck(L, tris, ro, rd)
    assert r["errors"] == 0 and r["mismatch"] == 0


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 64])
def test_product_bvh_tiny_and_degenerate(n):
    L = _hostcheck()
    rng = np.random.default_rng(n)
    tris = rng.uniform(0, 1, (max(n, 1), 9)).astype(np.float32)[:n]
    ro = rng.uniform(-1, 2, (200, 3)).astype(np.float32)
    rd = rng.normal(size=(200, 3))
    rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    r = _selfcheck(L, tris.reshape(-1, 9), ro, rd)
    assert r["errors"] == 0 and r["mismatch"] == 0
    # coincident triangles (SAH finds no plane): must still terminate with bounded leaves
    if n >= 5:
        same = np.repeat(tris[:1], n, axis=0)
        r = _selfcheck(L, same, ro, rd)
        assert r["errors"] == 0 and r["mismatch"] == 0 and r["maxleaf"] <= 4


# ----------------------------------------------------------------------------- sharding
def test_shard_ranges_partition_the_slots():
    W = rtdist.W
    for world in (1, 2, 4, 8, 64):
        ranges = [rtdist.shard_range(r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == W
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    with pytest.raises(ValueError):
        rtdist.shard_range(0, 3)
    with pytest.raises(ValueError):
        rtdist.shard_range(8, 8)


def test_camera_ray_ownership_is_a_round_robin_of_pixel_strips():
    W, spp, world = rtdist.W, 256, 8
    # all samples of a pixel go to one rank, consecutive 512-pixel strips cycle over the ranks
    per_rank_pixels = W // world // spp
    assert per_rank_pixels == 512
    for pixel in (0, 511, 512, 4095, 4096, 2073599):
        owners = {rtdist.owner_of_camera_ray(pixel * spp + s, world) for s in (0, 1, spp - 1)}
        assert len(owners) == 1
        assert owners.pop() == (pixel // per_rank_pixels) % world


def test_full_pool_k_paths_build_does_not_spill_vector_registers():
    """The 4-waves-per-SIMD builds of k_paths live on a 128-VGPR budget; source changes that tip the register allocator
    into spilling cost 4 - 6 % and look like noise in a benchmark (a spill inside the node loop: far more).  hipcc
    cross-compiles without a GPU: its resource remarks must report no VGPR spill for the bench configuration (4-wide
    nodes) and for the 2-wide alternative."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) and not shutil.which("hipcc"):
        pytest.skip("no hipcc in this environment")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "rtcuda_amd", "csrc"), "resource-usage"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = r.stdout.splitlines()
    found = 0
    # k_paths<LDS_TABLES = true, WIDE = true (bench) / false, MAJORITY = true, MIN_WAVES = 4, DRAW_CIDS, LITERAL, VERIFY>:
    #   (0, 0, 1) the DEFAULT build -- the one bench.py times: the reference's decisions on the product's walk;
    #   (0, 0, 0) RT_FLAG_WATERTIGHT;  (1, 0, 0) the per-sample-RNG build (fewer registers: at least 4 waves);
    #   (0, 1, 0) RT_FLAG_REFERENCE_WALK (round 4: 53 spilled registers -- its private stack had been promoted to 32 VGPRs;
    #   since round 5 the walk borrows the lane's LDS stack column)
    builds = [(0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0)]
    for k, line in enumerate(lines):
        for draw, lit, ver in builds:
            for wide in (1, 0):
                if lit and wide:
                    continue  # (the literal walk does not look at the product's node format: one build)
                if f"Function Name: _Z7k_pathsILb1ELb{wide}ELb1ELi4ELb{draw}ELb{lit}ELb{ver}EE" in line:
                    block = "\n".join(lines[k:k + 12])
                    m_spill = re.search(r"VGPRs Spill: (\d+)", block)
                    m_occ = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", block)
                    assert m_spill and m_occ, block
                    assert int(m_occ.group(1)) >= 4, block
                    assert int(m_spill.group(1)) == 0, block
                    found += 1
    assert found == 7, f"{found} of the 7 full-pool k_paths builds found in the resource remarks"


def test_bench_refuses_debug_flags_without_allow_invalid():
    """--debug-flags changes what the kernels do (0x100 drops the framebuffer deposits): bench.py must not produce a
    number from such a run unless told to, and then marks the line."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--debug-flags", "256"], capture_output=True, text=True)
    assert r.returncode != 0 and "allow-invalid" in (r.stderr + r.stdout)


def test_bvh_optimisation_keeps_every_hit_and_saves_node_steps(oracle, bunny_matte):
    """rt_bvh.h optimize_reinsert: the insertion-based pass after the SAH sweep changes the topology only.  Same closest
    hits (triangle and t) on the CPU walk with and without it, fewer node steps per ray with it."""
    import ctypes
    import subprocess
    import sys
    code = r'''
import ctypes, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from rtcuda_amd import scenes
import raygen
from oracle.oracle import Oracle
orc = Oracle("pinned")
a = scenes.cornell_bunny("matte")
cam = orc.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, 16 / 9)
o, d = raygen.camera_rays(cam, 1920, 1080, 60000, seed=77)
H = ctypes.CDLL(%r)
H.rt_hostwalk_create.restype = ctypes.c_void_p
H.rt_hostwalk_create.argtypes = [ctypes.c_void_p, ctypes.c_int]
H.rt_hostwalk_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6
H.rt_hostwalk_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
tris = np.ascontiguousarray(a.tris, np.float32)
h = H.rt_hostwalk_create(tris.ctypes.data, len(tris))
n = len(o); tri = np.zeros(n, np.int32); t = np.zeros(n, np.float32); tm = np.full(n, 3.4028234663852886e38, np.float32)
st = np.zeros(4, np.int64)
H.rt_hostwalk_stats(st.ctypes.data, 1)
assert H.rt_hostwalk_trace(h, 0, n, o.ctypes.data, d.ctypes.data, tm.ctypes.data, None, tri.ctypes.data, t.ctypes.data) == 0
H.rt_hostwalk_stats(st.ctypes.data, 1)
np.savez(sys.argv[1], tri=tri, t=t, nodes=st[1], rays=st[0])
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "rtcuda_amd", "librt_hostcheck.so"))
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for passes in ("0", "2"):  # (the pass count is read once per process: one process per setting)
            path = os.path.join(td, f"walk{passes}.npz")
            subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, RT_BVH_OPT=passes))
            out[passes] = dict(np.load(path))
    assert np.array_equal(out["0"]["tri"], out["2"]["tri"]) and np.array_equal(out["0"]["t"], out["2"]["t"])
    assert (out["0"]["tri"] >= 0).mean() > 0.4
    assert out["2"]["nodes"] < 0.97 * out["0"]["nodes"]  # measured: -7 % on these primary rays, -10 % over whole paths


# ----------------------------------------------------------------------------- RT_FLAG_REFERENCE_WALK: the reference's tree
def _ref_tree(L, tris):
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    n = tris.shape[0]
    L.rt_ref_tree_export.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5
    depth = ctypes.c_int(0)
    nn = L.rt_ref_tree_export(tris.ctypes.data, n, None, None, None, None, ctypes.byref(depth))
    bounds = np.zeros((nn, 6), np.float32)
    count, link, prims = np.zeros(nn, np.int32), np.zeros(nn, np.int32), np.zeros(max(n, 1), np.int32)
    L.rt_ref_tree_export(tris.ctypes.data, n, bounds.ctypes.data, count.ctypes.data, link.ctypes.data, prims.ctypes.data,
                         ctypes.byref(depth))
    return bounds, count, link, prims[:n], depth.value


@pytest.mark.parametrize("variant", ["matte", "sixteen_lights", "four_bunnies"])
def test_reference_tree_of_the_product_is_the_oracles_node_for_node(oracle, variant):
    """The tree RT_FLAG_REFERENCE_WALK walks (rt_ref_tree.h, the product's own builder after bvh.cuh:30-219) against the
    oracle's restatement of the same constructor: every node's bounds, primitive count and child / first-primitive index,
    and the primitive order -- bit for bit -- plus the BVH statistics SURVEY Appendix C records for these scenes."""
    from rtcuda_amd import scenes
    arrays = scenes.cornell_bunny(variant)
    L = _hostcheck()
    bounds, count, link, prims, depth = _ref_tree(L, arrays.tris)
    ob, oc, oi, op = oracle.scene(arrays).nodes()
    assert bounds.shape == ob.shape and np.array_equal(bounds.view(np.uint32), ob.view(np.uint32))
    assert np.array_equal(count, oc) and np.array_equal(link, oi) and np.array_equal(prims, op)
    expect = {"matte": (75687, 69463, 20), "sixteen_lights": (75701, 69477, 20), "four_bunnies": (302715, 277816, 22)}[variant]
    assert (bounds.shape[0], prims.shape[0], depth) == expect  # "BVH has N nodes and M primitives, with max_depth = D"


@pytest.mark.parametrize("variant", ["matte", "sixteen_lights", "four_bunnies"])
def test_reference_tree_boxes_are_nested_exactly(variant):
    """The premise of the default kernels' visibility test (rtcuda_amd.hip: ref_visible): along every
    root-to-leaf path of the reference's tree the boxes are NESTED EXACTLY -- a child's bounds lie inside its parent's with no
    rounding in between (both are min / max of the same fp32 triangle bounds: bvh.cuh:57-61,150-160) -- and every leaf's box
    contains the own box of each of its triangles as triangle.cuh:22-37 computes it from the stored record (p1 = p0 - e1,
    p2 = p0 + e2).  With fp32 rounding monotone, that is all "the leaf's box passes => every ancestor's passes" needs."""
    from rtcuda_amd import scenes
    arrays = scenes.cornell_bunny(variant)
    bounds, count, link, prims, _ = _ref_tree(_hostcheck(), arrays.tris)
    inner = np.where(count == 0)[0]
    for child in (link[inner], link[inner] + 1):
        assert (bounds[child][:, 0::2] >= bounds[inner][:, 0::2]).all()   # mins
        assert (bounds[child][:, 1::2] <= bounds[inner][:, 1::2]).all()   # maxs
    t = np.ascontiguousarray(arrays.tris, np.float32).reshape(-1, 3, 3)
    p0 = t[:, 0]
    e1, e2 = p0 - t[:, 1], t[:, 2] - p0
    p1, p2 = p0 - e1, p0 + e2                                              # the reference's reconstructions, fp32
    lo = np.minimum(p0, np.minimum(p1, p2))
    hi = np.maximum(p0, np.maximum(p1, p2))
    leaves = np.where(count > 0)[0]
    seen = np.zeros(len(t), bool)
    for k in leaves:
        idx = prims[link[k]:link[k] + count[k]]
        seen[idx] = True
        assert (lo[idx] >= bounds[k][0::2]).all() and (hi[idx] <= bounds[k][1::2]).all(), k
        # ... and is exactly their union
        assert np.array_equal(lo[idx].min(axis=0), bounds[k][0::2]) and np.array_equal(hi[idx].max(axis=0), bounds[k][1::2])
    assert seen.all()


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 200])
def test_reference_tree_tiny_and_coincident(oracle, n):
    """Small and degenerate inputs (equal centres: the order std::sort leaves among equal keys is part of the tree)."""
    from rtcuda_amd.scenes import SceneArrays
    rng = np.random.default_rng(100 + n)
    tris = rng.uniform(0, 1, (max(n, 1), 9)).astype(np.float32)[:n]
    if n >= 7:
        tris[3] = tris[1]          # coincident triangles
        tris[5, 0::3] = tris[2, 0::3]  # equal x coordinates: ties on one axis only
    L = _hostcheck()
    bounds, count, link, prims, depth = _ref_tree(L, tris)
    if n == 0:
        assert bounds.shape[0] == 1 and count[0] == 0
        return
    from rtcuda_amd import scenes
    base = scenes.cornell_bunny("matte", bunny=False)
    arrays = SceneArrays(tris=tris, tri_material=np.zeros(n, np.int32), tri_light=np.full(n, -1, np.int32),
                         materials=base.materials, lights=base.lights[:0])
    ob, oc, oi, op = oracle.scene(arrays).nodes()
    assert np.array_equal(bounds.view(np.uint32), ob.view(np.uint32))
    assert np.array_equal(count, oc) and np.array_equal(link, oi) and np.array_equal(prims, op)
