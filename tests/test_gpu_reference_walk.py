"""The reference's own decisions on the GPU against the LITERAL oracle.  Run with -m gpu.

Two results of lashhw/rtcuda are properties of its own tree, not of the scene: which accepted hits its fp32 slab test loses
(aabb_intersector.cuh:14-36, about 1 ray in 10^7) and which of two hits at exactly equal t wins (triangle.cuh:49: the later
tested one).  Two ways to the same answers are held here, side by side:
  * RT_FLAG_REFERENCE_WALK: the library builds the reference's tree (rt_ref_tree.h after bvh.cuh:30-219; held node for node
    against the oracle's on the CPU: tests/test_host_logic.py) and every ray walks it as Bvh::traverse does (bvh.cuh:221-357);
  * the DEFAULT kernels: the product's own walk, with every hit checked against what the reference's walk can see
    (ref_visible) and the rare rest re-traced literally.
Bar: EXACT equality with the oracle's literal mode -- every ray's triangle, t, u, v and occlusion flag bit for bit, every
integer event total of every frame up to the six full BASELINE frames, images within the float-atomics noise (RMS < 2e-6),
fixed-point sums equal to the oracle's and between shardings.
"""
import json
import os

import numpy as np
import pytest

from conftest import default_camera, oracle_render, oracle_scene, usable_cpus

pytestmark = pytest.mark.gpu

W = 1 << 20
FLT_MAX = np.float32(3.4028234663852886e38)
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = np.load(os.path.join(HERE, "golden", "render_goldens.npz"))
EVENTS = (("shade_events", "sum_mat"), ("any_rays", "sum_ah"), ("emission_adds", "emission_adds"),
          ("shadow_adds", "ah_adds"), ("rr_draws", "rr_draws"))


@pytest.fixture(scope="module")
def api():
    from rtcuda_amd import api as _api
    _api.lib()
    return _api


_scene_cache = {}


def _gpu_scene(api, variant):
    if variant not in _scene_cache:
        from rtcuda_amd import scenes
        _scene_cache[variant] = api.Scene(scenes.cornell_bunny(variant))
    return _scene_cache[variant]


def _rms(a, b):
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb)
    d = np.where(na, 0.0, a.astype(np.float64) - np.where(nb, 0.0, b.astype(np.float64)))
    return float(np.sqrt(np.mean(d ** 2)))


@pytest.mark.parametrize("how", ["reference_walk", "default"])
def test_every_ray_of_a_literal_oracle_render_comes_back_with_the_references_answer(api, oracle, bunny_matte, how):
    """Ray by ray: all path rays and shadow rays of a literal-oracle render of the matte scene, the ray the reference's walk
    is known to get wrong, rays along the axes (zero direction components: the FLT_EPSILON clamp of the slab set-up; also
    NEGATIVE zeros, for which the reference's octant and the sign of its 1 / d disagree and nearly every box fails) and
    rays that start on the walls (entry = exit on flat boxes) through rt_trace_closest_flags / rt_trace_any_flags --
    with RT_FLAG_REFERENCE_WALK (every ray through the reference's own tree) and with the DEFAULT kernels (the product's walk,
    ref_visible, the rare re-trace): triangle index, t, u, v and the occlusion flag equal the literal oracle's bit for bit --
    ties (the later tested triangle wins) and lost hits included."""
    import raygen
    F = api.FLAG_REFERENCE_WALK if how == "reference_walk" else 0
    sc = oracle.scene(bunny_matte)  # literal mode (default)
    oracle.raylog_enable(True)
    sc.render(default_camera(oracle, 1.0), 160, 160, 8, threads=usable_cpus())
    log = oracle.raylog_fetch()
    oracle.raylog_enable(False)
    # path ray 1 836 499 of the matte 256 x 256 x 40 frame: the reference returns wall 69458 behind light 69462
    miss_o = np.array([1052665855, 1062038174, 3212836608], np.uint32).view(np.float32).reshape(1, 3)
    miss_d = np.array([1048527110, 1054521025, 1063157561], np.uint32).view(np.float32).reshape(1, 3)
    ao, ad = raygen.axis_aligned_rays(4000, seed=11)
    rng = np.random.default_rng(5)
    nz_o, nz_d = raygen.axis_aligned_rays(3000, seed=12)
    nz_d = np.where(nz_d == 0, np.float32(-0.0), nz_d)  # -0.0 components (np.where keeps the sign bit)
    assert np.signbit(nz_d[nz_d == 0]).all() and (nz_d == 0).any()
    ao, ad = np.concatenate([ao, nz_o]), np.concatenate([ad, nz_d])
    wall_o = rng.uniform(0, 1, (4000, 3)).astype(np.float32)
    wall_o[np.arange(4000), rng.integers(0, 3, 4000)] = rng.choice(np.array([0.0, 1.0, -1.0], np.float32), 4000)
    wall_d = rng.normal(size=(4000, 3))
    wall_d = (wall_d / np.linalg.norm(wall_d, axis=1, keepdims=True)).astype(np.float32)
    o = np.concatenate([log["closest_o"], miss_o, ao, wall_o])
    d = np.concatenate([log["closest_d"], miss_d, ad, wall_d])
    tmax = np.full(len(o), FLT_MAX, np.float32)
    gpu = _gpu_scene(api, "matte")
    tri, t, u, v = gpu.trace_closest(o, d, tmax, flags=F)
    wt, tt, uu, vv = sc.trace_closest(o, d, tmax, threads=usable_cpus())
    n_log = len(log["closest_tri"])
    assert np.array_equal(wt[:n_log], log["closest_tri"])          # (the log and the stage entry point agree)
    assert int(wt[n_log]) == 69458 and int(tri[n_log]) == 69458      # the lost hit is lost here too
    assert np.array_equal(tri, wt)
    hit = wt >= 0
    for got, want in ((t, tt), (u, uu), (v, vv)):
        assert np.array_equal(got[hit].view(np.uint32), want[hit].view(np.uint32))
    # ... and RT_FLAG_WATERTIGHT finds the light the reference loses on that ray
    tri_d, _, _, _ = gpu.trace_closest(miss_o, miss_d, np.full(1, FLT_MAX, np.float32), flags=api.FLAG_WATERTIGHT)
    assert int(tri_d[0]) == 69462
    occ = gpu.trace_any(log["any_o"], log["any_d"], log["any_tmax"], log["any_excluded"], flags=F)
    assert np.array_equal(occ, log["any_occluded"])
    tm2 = rng.uniform(0.05, 2.0, len(ao) + len(wall_o)).astype(np.float32)
    excl = rng.integers(-1, bunny_matte.n_tris, len(tm2)).astype(np.int32)
    o2, d2 = np.concatenate([ao, wall_o]), np.concatenate([ad, wall_d])
    assert np.array_equal(gpu.trace_any(o2, d2, tm2, excl, flags=F), sc.trace_any(o2, d2, tm2, excl, threads=usable_cpus()))
    assert n_log > 500_000 and len(occ) > 200_000


CASES = [
    # variant, w, h, spp, max_bounces, seed
    ("matte", 64, 64, 4, 10, 1),              # one generation: lockstep rounds only (k_advance + k_trace<literal>)
    ("matte", 256, 256, 4, 10, 1),            # BASELINE config 1's workload
    ("matte", 256, 256, 40, 10, 1),           # 2.5 generations: the frame with the audited lost hit
    ("full_bsdf", 300, 200, 48, 10, 1),       # glass + mirror, spp does not divide W, one NaN pixel
    ("full_bsdf", 400, 300, 20, 20, 12345),   # long Russian-roulette chains, another seed
    ("four_bunnies", 480, 270, 20, 10, 1),    # the deep tree (depth 22)
    ("sixteen_lights", 480, 270, 48, 10, 1),  # the scene whose flat light boxes the reference's slab test drops hits on
]


@pytest.mark.parametrize("how", ["reference_walk", "default"])
@pytest.mark.parametrize("variant,w,h,spp,max_bounces,seed", CASES)
def test_reference_walk_frame_equals_the_literal_oracle(api, oracle, variant, w, h, spp, max_bounces, seed, how):
    """Whole frames against the LITERAL oracle: with RT_FLAG_REFERENCE_WALK and with the default kernels."""
    img_c, _, st_c = oracle_render(oracle, variant, w, h, spp, max_bounces=max_bounces, seed=seed, watertight=False)
    gpu = _gpu_scene(api, variant)
    img_g, st_g = gpu.render(api.make_camera(aspect=w / h), w, h, spp, max_bounces=max_bounces, seed=seed,
                             flags=api.FLAG_REFERENCE_WALK if how == "reference_walk" else 0)
    assert st_g["camera_rays"] == w * h * spp
    for kg, kc in EVENTS:
        assert st_g[kg] == st_c[kc], (kg, st_g[kg], st_c[kc])
    assert _rms(img_g, img_c) < 2e-6
    assert np.nan_to_num(np.abs(img_g.astype(np.float64) - img_c)).max() < 1e-4


def test_reference_walk_differs_from_the_watertight_walk_where_the_audit_says(api, oracle):
    """The sixteen-light 480 x 270 x 48 frame: the reference's walk leaves 7 shadow rays unoccluded that exhaustive search --
    and RT_FLAG_WATERTIGHT -- find occluded.  All three modes on the GPU, each equal to its oracle mode;
    the default kernels' counter of lost hits says 7."""
    w, h, spp = 480, 270, 48
    gpu = _gpu_scene(api, "sixteen_lights")
    cam = api.make_camera(aspect=w / h)
    _, st_ref = gpu.render(cam, w, h, spp, flags=api.FLAG_REFERENCE_WALK)
    _, st_def = gpu.render(cam, w, h, spp)
    _, st_wat = gpu.render(cam, w, h, spp, flags=api.FLAG_WATERTIGHT)
    _, _, lit = oracle_render(oracle, "sixteen_lights", w, h, spp, watertight=False)
    _, _, wat = oracle_render(oracle, "sixteen_lights", w, h, spp, watertight=True)
    assert [st_ref[g] for g, _ in EVENTS] == [lit[c] for _, c in EVENTS]
    assert [st_def[g] for g, _ in EVENTS] == [lit[c] for _, c in EVENTS]
    assert [st_wat[g] for g, _ in EVENTS] == [wat[c] for _, c in EVENTS]
    assert st_def["reference_lost_hits"] == 7, st_def
    assert [lit[c] for _, c in EVENTS] != [wat[c] for _, c in EVENTS]  # (the two definitions do differ on this frame)


@pytest.mark.parametrize("variant,w,h,spp", [("matte", 160, 100, 160), ("full_bsdf", 128, 72, 256)])
def test_reference_walk_reproduces_the_committed_literal_fixtures(api, variant, w, h, spp):
    """tests/golden/render_goldens.npz `literal_*`: outputs of the literal oracle (tests/golden/make_golden.py).  The reference
    walk must reproduce the counts EXACTLY (and so must the default kernels: tests/test_gpu_multigen.py)."""
    key = f"literal_{variant}_{w}x{h}x{spp}"
    gpu = _gpu_scene(api, variant)
    img, st = gpu.render(api.make_camera(aspect=w / h), w, h, spp, flags=api.FLAG_REFERENCE_WALK)
    got = [st["shade_events"], st["any_rays"], st["emission_adds"], st["shadow_adds"], st["rr_draws"], st["camera_rays"]]
    assert got == GOLDEN[key + "_counts"].tolist()
    assert _rms(img, GOLDEN[key + "_img"].astype(np.float32)) < 2e-6


def test_reference_walk_is_partition_invariant_and_bit_reproducible(api):
    """RT_FLAG_REFERENCE_WALK | RT_FLAG_DETERMINISTIC: the int64 sums of 8 slot-range shards (the small-shard build of the
    literal kernel) equal the unsharded frame's, twice."""
    import torch
    w, h, spp = 300, 200, 48
    gpu = _gpu_scene(api, "full_bsdf")
    cam = api.make_camera(aspect=w / h)
    F = api.FLAG_REFERENCE_WALK
    full = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    st_full = gpu.render_shard_fixed(cam, w, h, spp, 0, 1, full.data_ptr(), flags=F)
    again = torch.zeros_like(full)
    gpu.render_shard_fixed(cam, w, h, spp, 0, 1, again.data_ptr(), flags=F)
    acc = torch.zeros_like(full)
    keys = ("camera_rays", "shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws")
    tot = {k: 0 for k in keys}
    for r in range(8):
        st = gpu.render_shard_fixed(cam, w, h, spp, r, 8, acc.data_ptr(), flags=F)
        for k in keys:
            tot[k] += st[k]
    torch.cuda.synchronize()
    assert torch.equal(full, again) and torch.equal(acc, full)
    assert tot == {k: st_full[k] for k in keys}
    with pytest.raises(api.RtError):  # a parity mode and a non-parity mode do not combine
        gpu.render_shard_fixed(cam, w, h, spp, 0, 1, acc.data_ptr(), flags=F | api.FLAG_RNG_PER_SAMPLE)


def _full_size_frames():
    return json.load(open(os.path.join(HERE, "golden", "full_size_event_totals.json")))["frames"]


@pytest.mark.parametrize("frame", _full_size_frames(), ids=lambda f: f"{f['scene']}_{f['width']}x{f['height']}x{f['spp']}")
def test_every_full_baseline_frame_equals_the_literal_oracle_under_the_reference_walk(api, frame):
    """The six full BASELINE frames under RT_FLAG_REFERENCE_WALK: the five integer event totals EQUAL the committed
    `oracle_literal` column (tests/golden/full_size_event_totals.json) -- as the default kernels do (tests/test_gpu_multigen.py)."""
    import torch
    w, h, spp = frame["width"], frame["height"], frame["spp"]
    gpu = _gpu_scene(api, frame["scene"])
    fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    st = gpu.render_shard(api.make_camera(aspect=w / h), w, h, spp, 0, 1, fb.data_ptr(), flags=api.FLAG_REFERENCE_WALK)
    torch.cuda.synchronize()
    assert st["camera_rays"] == frame["samples"]
    for k, v in frame["oracle_literal"].items():
        assert st[k] == v, (k, st[k], v)
    if "oracle_literal_nan_pixels" in frame:
        nan_pixels = int(torch.isnan(fb).view(-1, 3).any(dim=1).sum().item())
        assert nan_pixels == frame["oracle_literal_nan_pixels"]


def _tiny_scenes():
    """(name, arrays): the bare Cornell box (12 triangles: a tree of depth 4) with a point light only / one area light + one
    point light / no light at all, one single triangle (the root is a leaf: bvh.cuh:252,307), and no triangle at all."""
    from rtcuda_amd import scenes
    out = []
    for kind in ("point", "mixed", "none"):
        a = scenes.cornell_bunny("matte", bunny=False)
        if kind == "point":
            lights = np.zeros(1, scenes.LIGHT_DTYPE)
            lights[0] = (scenes.POINT_LIGHT, (0.7, 0.15, -0.6), -1, (0.5, 0.5, 0.5))
            a.lights, a.tri_light = lights, np.full(a.n_tris, -1, np.int32)
        elif kind == "mixed":
            lights = np.zeros(2, scenes.LIGHT_DTYPE)
            lights[0] = a.lights[0]
            lights[1] = (scenes.POINT_LIGHT, (0.3, 0.8, -0.3), -1, (0.2, 0.3, 0.4))
            a.lights, a.tri_light = lights, np.where(a.tri_light == 0, 0, -1).astype(np.int32)
        else:
            a.lights, a.tri_light = np.zeros(0, scenes.LIGHT_DTYPE), np.full(a.n_tris, -1, np.int32)
        out.append((kind, a))
    base = scenes.cornell_bunny("matte", bunny=False)
    out.append(("one_triangle", scenes.SceneArrays(tris=base.tris[8:9].copy(), tri_material=np.array([2], np.int32),
                                                   tri_light=np.array([-1], np.int32), materials=base.materials,
                                                   lights=np.zeros(0, scenes.LIGHT_DTYPE))))
    out.append(("empty", scenes.SceneArrays(tris=np.zeros((0, 9), np.float32), tri_material=np.zeros(0, np.int32),
                                            tri_light=np.zeros(0, np.int32), materials=base.materials,
                                            lights=np.zeros(0, scenes.LIGHT_DTYPE))))
    return out


@pytest.mark.parametrize("name,arrays", _tiny_scenes(), ids=[n for n, _ in _tiny_scenes()])
def test_reference_walk_on_tiny_and_degenerate_scenes(api, oracle, name, arrays):
    """The reference's tree at its edges -- a root that is a leaf, a tree of a dozen triangles, no triangle at all -- and the
    light code paths (point light: no excluded triangle; no lights: no shadow rays) under RT_FLAG_REFERENCE_WALK, alone and
    through rt_render_multi: event totals and fixed-point sums equal to the literal oracle's."""
    import torch
    w, h, spp = 48, 48, 8
    want = np.zeros((h, w, 3), np.int64)
    _, _, st_c = oracle.scene(arrays).render(default_camera(oracle, 1.0), w, h, spp, threads=usable_cpus(), fixed_out=want)
    gpu = api.Scene(arrays)
    cam = api.make_camera(aspect=1.0)
    got = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    st = gpu.render_shard_fixed(cam, w, h, spp, 0, 1, got.data_ptr(), flags=api.FLAG_REFERENCE_WALK)
    got_d = torch.zeros_like(got)
    st_d = gpu.render_shard_fixed(cam, w, h, spp, 0, 1, got_d.data_ptr())  # the default kernels on the same edge cases
    torch.cuda.synchronize()
    assert st["camera_rays"] == w * h * spp
    for kg, kc in EVENTS:
        assert st[kg] == st_c[kc] == st_d[kg], (name, kg, st[kg], st_c[kc], st_d[kg])
    assert np.array_equal(got.cpu().numpy().reshape(h, w, 3), want)
    assert torch.equal(got_d, got)
    img_m, st_m = gpu.render_multi(cam, w, h, spp, [0, 0], flags=api.FLAG_REFERENCE_WALK | api.FLAG_DETERMINISTIC)
    img_1, _ = gpu.render(cam, w, h, spp, flags=api.FLAG_REFERENCE_WALK | api.FLAG_DETERMINISTIC)
    assert all(st_m[kg] == st_c[kc] for kg, kc in EVENTS)
    assert np.array_equal(img_m.view(np.uint32), img_1.view(np.uint32))
    if name in ("none", "empty"):
        assert st["any_rays"] == 0 and not img_1.any()
    gpu.close()
