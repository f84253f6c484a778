"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle.  Run with -m gpu.

Every render in THIS file has at most W = 1 048 576 camera rays, i.e. ONE generation -- which is the final generation
and therefore runs on the lockstep round pipeline (k_advance + k_trace); the persistent kernel k_paths only parks the
slots.  What is covered here is that pipeline, the stage-level entry points, the host logic and the edge cases of the
final generation (tiny frames, termination rule).  The persistent kernel -- the one bench.py times -- is held against
the oracle over several generations in tests/test_gpu_multigen.py."""
import numpy as np
import pytest

from conftest import default_camera, usable_cpus
import raygen

pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(3.4028234663852886e38)


@pytest.fixture(scope="module")
def api():
    from rtcuda_amd import api as _api
    _api.lib()  # raises if the HIP library is missing: there is no fallback
    return _api


@pytest.fixture(scope="module")
def gpu_matte(api, bunny_matte):
    return api.Scene(bunny_matte)


@pytest.fixture(scope="module")
def gpu_full(api, bunny_full_bsdf):
    return api.Scene(bunny_full_bsdf)


@pytest.fixture(scope="module")
def cpu_matte(oracle, bunny_matte):
    return oracle.scene(bunny_matte)


@pytest.fixture(scope="module")
def cpu_full(oracle, bunny_full_bsdf):
    return oracle.scene(bunny_full_bsdf)


def test_xorwow_states_and_draws_bit_exact(api, oracle):
    for first, count in [(0, 4096), (1 << 19, 1024), ((1 << 20) - 2048, 2048), (123457, 777)]:
        st, uni = api.xorwow_states(1, first, count, draws=5)
        ref = oracle.xorwow_init_range(1, first, count)
        ref_u = np.zeros((count, 5), np.float32)
        for i in range(count):
            s = ref[i].copy()
            _, ref_u[i] = oracle.xorwow_draw(s, 5)
            ref[i] = s
        assert np.array_equal(st, ref)
        assert np.array_equal(uni.view(np.uint32), ref_u.view(np.uint32))
    st, _ = api.xorwow_states(0xDEADBEEF12345, 5, 16, draws=0)
    assert np.array_equal(st, oracle.xorwow_init_range(0xDEADBEEF12345, 5, 16))


def _closest_compare(gpu, cpu, o, d, tmax, max_mismatch_frac=0.0):
    """Default kernels against the LITERAL oracle.  Round 5: the default kernels make the reference's decisions (which hits
    its walk can see, who wins an exact tie), so EVERY ray must agree on the triangle; until round 4 a fraction of ties and
    lost hits was allowed here."""
    g = gpu.trace_closest(o, d, tmax)
    c = cpu.trace_closest(o, d, tmax)
    same_tri = g[0] == c[0]
    n = len(o)
    bad = np.where(~same_tri)[0]
    assert len(bad) <= max_mismatch_frac * n, f"{len(bad)} of {n} rays disagree on the hit triangle: {bad[:8]}"
    hit = same_tri & (c[0] >= 0)
    for k in (1, 2, 3):  # t, u, v bit for bit
        assert np.array_equal(g[k][hit].view(np.uint32), c[k][hit].view(np.uint32))
    return g, c, len(bad)


def test_trace_closest_matches_oracle(api, oracle, gpu_matte, cpu_matte):
    cam = default_camera(oracle, 16 / 9)
    o, d = raygen.camera_rays(cam, 1920, 1080, 400_000, seed=11)
    tmax = np.full(len(o), FLT_MAX, np.float32)
    g, c, nbad = _closest_compare(gpu_matte, cpu_matte, o, d, tmax)
    assert 0.4 < (c[0] >= 0).mean() < 0.7  # 53 % of 16:9 primary rays hit the box (SURVEY Appx C)
    o2, d2 = raygen.bounce_rays(o, d, c[1], c[0] >= 0, seed=12)
    _closest_compare(gpu_matte, cpu_matte, o2, d2, np.full(len(o2), FLT_MAX, np.float32))
    # finite tmax (shadow-ray style) and degenerate directions
    tm = np.random.default_rng(5).uniform(0.05, 1.5, len(o2)).astype(np.float32)
    _closest_compare(gpu_matte, cpu_matte, o2, d2, tm)
    o3, d3 = raygen.axis_aligned_rays(50_000, seed=13)
    _closest_compare(gpu_matte, cpu_matte, o3, d3, np.full(len(o3), FLT_MAX, np.float32))


def test_rays_that_start_far_outside_the_scene_match_the_oracle(api, oracle, bunny_matte):
    """The 4-wide records are padded for the ray origins that are traced (one-fma plane distance: rt_bvh.h); the hooks look at
    the origins of a batch and re-pad when they lie outside (ensure_origin_radius).  Rays from 3 to 60 scene sizes away, aimed at
    points near the vertices and edges of random triangles, on a scene object of its own (so that the padding starts at the
    scene's bounds): triangle, t, u, v as the literal oracle has them, closest and any hit -- and the same rays again after
    the widest batch."""
    gpu = api.Scene(bunny_matte)
    cpu = oracle.scene(bunny_matte)
    tris = np.asarray(bunny_matte.tris, np.float32).reshape(-1, 3, 3)
    rng = np.random.default_rng(31)
    batches = []
    for scale in (3.0, 20.0, 60.0, 3.0):
        n = 60_000
        tsel = tris[rng.integers(0, len(tris), n)]
        w = rng.dirichlet([0.3, 0.3, 0.3], n).astype(np.float32)
        target = (tsel * w[:, :, None]).sum(axis=1)
        dirs = rng.normal(size=(n, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        o = (target - dirs * scale * rng.uniform(0.3, 3.0, (n, 1))).astype(np.float32)
        d = target - o.astype(np.float64)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        batches.append((o, d))
    for o, d in batches:
        tmax = np.full(len(o), FLT_MAX, np.float32)
        g, c, _ = _closest_compare(gpu, cpu, o, d, tmax)
        assert (c[0] >= 0).mean() > 0.5
        tm = (0.999 * np.where(c[0] >= 0, c[1], 1.0)).astype(np.float32)  # shadow-ray style: up to just before the first hit
        excl = np.full(len(o), -1, np.int32)
        assert np.array_equal(gpu.trace_any(o, d, tm, excl), cpu.trace_any(o, d, tm, excl))
    gpu.close()


def test_trace_any_matches_oracle(api, oracle, gpu_matte, cpu_matte, bunny_matte):
    cam = default_camera(oracle, 1.0)
    o, d = raygen.camera_rays(cam, 512, 512, 100_000, seed=21)
    c = cpu_matte.trace_closest(o, d, np.full(len(o), FLT_MAX, np.float32))
    o2, d2 = raygen.bounce_rays(o, d, c[1], c[0] >= 0, seed=22)
    rng = np.random.default_rng(23)
    tm = rng.uniform(0.05, 1.2, len(o2)).astype(np.float32)
    light_tris = np.where(bunny_matte.tri_light >= 0)[0]
    excl = rng.choice(np.concatenate([light_tris, [-1]]), len(o2)).astype(np.int32)
    g_occ = gpu_matte.trace_any(o2, d2, tm, excl)
    c_occ = cpu_matte.trace_any(o2, d2, tm, excl)
    assert np.array_equal(g_occ, c_occ)
    assert 0.05 < c_occ.mean() < 0.95
    # empty batch is legal
    assert len(gpu_matte.trace_any(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32),
                                   np.zeros(0, np.float32), np.zeros(0, np.int32))) == 0


# Path ray 1 836 499 of the matte 256 x 256 x 40 frame (bit patterns): the reference's walk returns wall triangle 69458
# behind light triangle 69462 (tests/test_traversal_audit.py pins it on the CPU twin of the product's walk)
KNOWN_MISS_O = np.array([1052665855, 1062038174, 3212836608], np.uint32).view(np.float32).reshape(1, 3)
KNOWN_MISS_D = np.array([1048527110, 1054521025, 1063157561], np.uint32).view(np.float32).reshape(1, 3)
_raylog = {}


def _oracle_ray_log(oracle, bunny_matte, watertight):
    """Every ray of a 160 x 160 x 8 oracle render of the matte scene (~0.8 M rays) in the given mode, with the oracle's
    results; `miss_*`: what that mode answers on the ray the reference's walk gets wrong."""
    if watertight not in _raylog:
        sc = oracle.scene(bunny_matte).set_watertight(watertight)
        oracle.raylog_enable(True)
        sc.render(default_camera(oracle, 1.0), 160, 160, 8, threads=usable_cpus())
        log = oracle.raylog_fetch()
        oracle.raylog_enable(False)
        one = (sc.trace_closest_brute if watertight else sc.trace_closest)(KNOWN_MISS_O, KNOWN_MISS_D, np.full(1, FLT_MAX, np.float32))
        log["miss_tri"], log["miss_t"] = int(one[0][0]), one[1][0]
        _raylog[watertight] = log
    return _raylog[watertight]


@pytest.mark.parametrize("watertight", [False, True], ids=["default-vs-literal", "watertight-flag-vs-watertight"])
@pytest.mark.parametrize("env", [{}, {"RT_BVH_WIDE": "0"}, {"RT_STACK_CAP": "2"}, {"RT_BVH_WIDE": "0", "RT_STACK_CAP": "2"}],
                         ids=["wide", "pairs", "wide-overflow", "pairs-overflow"])
def test_oracle_ray_log_replayed_ray_by_ray(api, oracle, bunny_matte, monkeypatch, env, watertight):
    """The GPU arithmetic itself (v_rcp_f32 for 1/d, the one-comparison packed box test, the overflow stack; in the default
    build also ref_visible and the rare literal re-trace), ray by ray: every path and shadow ray of an oracle render goes
    through rt_trace_closest / rt_trace_any and must come back with the oracle's answer -- triangle index and t bit for bit,
    ties included, occlusion flag -- for both node formats and with the traversal stack forced through its global overflow
    part: the DEFAULT kernels against the LITERAL oracle, RT_FLAG_WATERTIGHT against the watertight oracle.  The ray the
    reference's own walk gets wrong (KNOWN_MISS) is one of them: wall 69458 behind the light by default, as in the
    reference; light 69462 with RT_FLAG_WATERTIGHT, as exhaustive search has it."""
    log = _oracle_ray_log(oracle, bunny_matte, watertight)
    flags = api.FLAG_WATERTIGHT if watertight else 0
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    gpu = api.Scene(bunny_matte)  # (RT_BVH_WIDE is read at scene creation)
    o = np.concatenate([log["closest_o"], KNOWN_MISS_O])
    d = np.concatenate([log["closest_d"], KNOWN_MISS_D])
    tri, t, _, _ = gpu.trace_closest(o, d, np.full(len(o), FLT_MAX, np.float32), flags=flags)
    want_tri = np.concatenate([log["closest_tri"], [log["miss_tri"]]])
    want_t = np.concatenate([log["closest_t"], [log["miss_t"]]])
    assert log["miss_tri"] == (69462 if watertight else 69458)
    assert np.array_equal(tri, want_tri)
    hit = want_tri >= 0
    assert np.array_equal(t[hit].view(np.uint32), want_t[hit].astype(np.float32).view(np.uint32))
    occ = gpu.trace_any(log["any_o"], log["any_d"], log["any_tmax"], log["any_excluded"], flags=flags)
    assert np.array_equal(occ, log["any_occluded"])
    gpu.close()
    assert len(o) > 500_000 and len(occ) > 200_000


def _rms(a, b):
    return np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=(0, 1)))


# (256 x 256 x 4: BASELINE config 1's exact workload -- "bun_zipper.ply, 256x256, 4 spp, diffuse-only" -- on the HIP path)
@pytest.mark.parametrize("w,h,spp", [(64, 36, 8), (64, 64, 4), (32, 32, 16), (256, 256, 4)])
def test_render_matches_oracle_matte(api, oracle, gpu_matte, cpu_matte, w, h, spp):
    cam = default_camera(oracle, w / h)
    img_c, sum_c, st_c = cpu_matte.render(cam, w, h, spp, threads=8)
    img_g, st_g = gpu_matte.render(api.make_camera(aspect=w / h), w, h, spp)
    # event counts are integers: exact
    assert st_g["camera_rays"] == w * h * spp
    assert st_g["shade_events"] == st_c["sum_mat"]
    assert st_g["any_rays"] == st_c["sum_ah"]
    assert st_g["emission_adds"] == st_c["emission_adds"]
    assert st_g["shadow_adds"] == st_c["ah_adds"]
    assert st_g["rr_draws"] == st_c["rr_draws"]
    # pixels: same contributions, only the float summation order differs (atomics)
    rms = _rms(img_g, img_c)
    assert rms.max() < 2e-6, rms
    assert np.abs(img_g - img_c).max() < 1e-4


@pytest.mark.parametrize("w,h,spp,max_bounces,seed", [
    # (one generation each: gen() runs once per slot, in k_advance; the per-generation pixel stepping and the
    #  non-lockstep Russian-roulette loop are NOT reached here -- see tests/test_gpu_multigen.py for those)
    (37, 23, 16, 10, 1),    # odd sizes
    (61, 7, 64, 10, 1),     # a very wide image
    (33, 19, 48, 10, 1),    # spp does not divide W: pixel = camera ray / spp by the 64-bit divide
    (40, 30, 3, 10, 7),     # another seed, odd spp
    (32, 24, 8, 0, 1),      # no bounce at all: emission only
    (32, 24, 8, 1, 1),
    (32, 24, 8, 2, 1),
    (24, 16, 8, 20, 1),     # Russian roulette over 20 lockstep rounds (two chunks of 16 enqueued rounds)
    (24, 16, 8, 100, 1),    # 102 possible rounds: seven chunks, the host looks at the stop rule's counter between them
])
def test_render_edge_configurations_match_oracle(api, oracle, gpu_full, cpu_full, w, h, spp, max_bounces, seed):
    """Sizes, sample counts, bounce limits and seeds through the lockstep pipeline (final-generation schedule, the
    reference's termination rule render.cuh:436): same integer event totals as the oracle, same image up to the float
    summation order."""
    cam = default_camera(oracle, w / h)
    img_c, sum_c, st_c = cpu_full.render(cam, w, h, spp, max_bounces=max_bounces, seed=seed, threads=8)
    img_g, st_g = gpu_full.render(api.make_camera(aspect=w / h), w, h, spp, max_bounces=max_bounces, seed=seed)
    assert st_g["camera_rays"] == w * h * spp
    assert st_g["shade_events"] == st_c["sum_mat"]
    assert st_g["any_rays"] == st_c["sum_ah"]
    assert st_g["emission_adds"] == st_c["emission_adds"]
    assert st_g["shadow_adds"] == st_c["ah_adds"]
    assert st_g["rr_draws"] == st_c["rr_draws"]
    assert _rms(img_g, img_c).max() < 2e-6
    assert np.abs(img_g - img_c).max() < 1e-4


def test_render_matches_oracle_full_bsdf(api, oracle, gpu_full, cpu_full):
    w, h, spp = 96, 54, 8
    cam = default_camera(oracle, w / h)
    img_c, _, st_c = cpu_full.render(cam, w, h, spp, threads=8)
    img_g, st_g = gpu_full.render(api.make_camera(aspect=w / h), w, h, spp)
    assert st_c["ch_adds"] == 0  # the BSDF-sampled MIS ray never contributes
    assert st_g["shade_events"] == st_c["sum_mat"]
    assert st_g["any_rays"] == st_c["sum_ah"]
    assert st_g["shadow_adds"] == st_c["ah_adds"]
    assert st_g["rr_draws"] == st_c["rr_draws"]
    assert _rms(img_g, img_c).max() < 2e-6


def test_camera_matches_oracle(api, oracle):
    for aspect in (1.0, 16 / 9, 0.5):
        assert np.array_equal(api.make_camera(aspect=aspect).view(np.uint32),
                              default_camera(oracle, aspect).view(np.uint32))


def test_shards_sum_to_full_image(api, oracle, gpu_matte):
    """Slot-range shards are disjoint in camera rays: their raw sums add up to the 1-GPU image."""
    import torch
    w, h, spp = 80, 45, 16
    cam = api.make_camera(aspect=w / h)
    full = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    st_full = gpu_matte.render_shard(cam, w, h, spp, 0, 1, full.data_ptr())
    for shards in (2, 8):
        acc = torch.zeros_like(full)
        rays = 0
        for r in range(shards):
            st = gpu_matte.render_shard(cam, w, h, spp, r, shards, acc.data_ptr())
            rays += st["camera_rays"]
        torch.cuda.synchronize()
        assert rays == st_full["camera_rays"] == w * h * spp
        a, b = acc.cpu().numpy(), full.cpu().numpy()
        assert np.allclose(a, b, rtol=2e-5, atol=1e-6)


def test_error_paths(api, bunny_matte):
    import copy
    bad = copy.copy(bunny_matte)
    bad.tri_material = bunny_matte.tri_material.copy()
    bad.tri_material[0] = 99
    with pytest.raises(api.RtError):
        api.Scene(bad)
    sc = api.Scene(bunny_matte)
    with pytest.raises(api.RtError):
        sc.render(api.make_camera(), 0, 10, 1)
    with pytest.raises(api.RtError):  # beyond the reference's int32 camera-ray range
        sc.render(api.make_camera(), 8192, 8192, 64)
    with pytest.raises(api.RtError):  # more pixels than 32-bit framebuffer indexing reaches (3 * pixel < 2^31): ADVICE r1
        sc.render(api.make_camera(), 32767, 32767, 1)


GOLDEN = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "render_goldens.npz"))


@pytest.mark.parametrize("variant,w,h,spp", [("matte", 32, 32, 16), ("full_bsdf", 48, 27, 8),
                                             ("sixteen_lights", 40, 30, 4), ("matte", 1, 1, 3), ("full_bsdf", 7, 5, 1)])
def test_render_matches_committed_goldens(api, variant, w, h, spp):
    """Against the committed fixtures (no oracle at run time): tests/golden/make_golden.py made them."""
    from rtcuda_amd import scenes
    key = f"{variant}_{w}x{h}x{spp}"
    sc = api.Scene(scenes.cornell_bunny(variant))
    img, st = sc.render(api.make_camera(aspect=w / h), w, h, spp, flags=api.FLAG_WATERTIGHT)  # (fixtures of the watertight oracle)
    counts = GOLDEN[key + "_counts"]
    assert [st["shade_events"], st["any_rays"], st["emission_adds"], st["shadow_adds"], st["rr_draws"],
            st["camera_rays"]] == counts.tolist()
    ref = GOLDEN[key + "_img"]
    assert _rms(img, ref).max() < 2e-6
    assert np.abs(img - ref).max() < 1e-4


def test_full_size_properties(api, gpu_full):
    """BASELINE-size frame (1920x1080) through size-independent properties: every camera ray is
    generated exactly once, shards partition the rays, the image is finite, non-negative and its
    mean matches the low-resolution oracle statistics; doubling the render does not change it."""
    w, h, spp = 1920, 1080, 4
    cam = api.make_camera(aspect=w / h)
    img, st = gpu_full.render(cam, w, h, spp)
    assert st["camera_rays"] == w * h * spp
    assert np.isfinite(img).all() and (img >= 0).all()
    # full-BSDF scene mean RGB at 480x270x4 = 0.12498 0.10625 0.08786 (same estimator)
    assert np.allclose(img.reshape(-1, 3).mean(0), [0.12498, 0.10625, 0.08786], atol=2e-3)
    img2, st2 = gpu_full.render(cam, w, h, spp)
    assert st2["shade_events"] == st["shade_events"] and st2["any_rays"] == st["any_rays"]
    assert _rms(img, img2).max() < 1e-6  # same paths; only the atomic summation order may differ


def test_cpp_drop_in_driver_matches_python_path(api, gpu_matte, tmp_path):
    """examples/cornell_bunny.cpp (the reference driver's recipe on include/rtcuda/rtcuda.hpp) renders the
    same PPM as the Python glue path: same C-ABI underneath, independent scene construction on top."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "cornell_bunny")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "rtcuda_amd", "csrc"), "example"])
    out = tmp_path / "image.ppm"
    w, h, spp = 96, 96, 4
    subprocess.check_call([exe, str(w), str(h), str(spp), os.path.join(ROOT, "data", "bun_zipper.ply"), str(out)],
                          cwd=ROOT)  # the executable links the system HIP runtime itself
    tok = out.read_text().split()
    assert tok[:4] == ["P3", str(w), str(h), "255"]
    got = np.array(tok[4:], np.int64).reshape(h, w, 3)
    img, _ = gpu_matte.render(api.make_camera(aspect=w / h), w, h, spp)
    ref = np.clip((np.float32(256.0) * img).astype(np.int64), 0, 255)
    # identical contributions; a float-atomic ordering difference can flip a quantisation boundary
    assert (got != ref).mean() < 1e-3 and np.abs(got - ref).max() <= 1


def test_single_generation_frame_is_the_same_with_and_without_the_persistent_launch(api, oracle, gpu_full, monkeypatch):
    """A one-generation frame: with the persistent launch (which only parks the slots here) and with RT_PERSISTENT=0 the
    lockstep pipeline produces the same totals and image.  The several-generation version of this comparison, where
    k_paths does the work, is test_gpu_multigen.test_round_pipeline_and_scheduling_variants_agree_over_generations."""
    w, h, spp = 160, 90, 16
    cam = api.make_camera(aspect=w / h)
    img_p, st_p = gpu_full.render(cam, w, h, spp)
    monkeypatch.setenv("RT_PERSISTENT", "0")
    img_r, st_r = gpu_full.render(cam, w, h, spp)
    monkeypatch.delenv("RT_PERSISTENT")
    for k in ("camera_rays", "shade_events", "closest_rays", "any_rays", "emission_adds", "shadow_adds", "rr_draws"):
        assert st_p[k] == st_r[k], k
    assert st_r["iterations"] > st_p["iterations"]
    assert _rms(img_p, img_r).max() < 2e-6


def test_both_node_formats_match_oracle(api, oracle, cpu_matte, bunny_matte, monkeypatch):
    """2-wide nodes (RT_BVH_WIDE=0) and 4-wide nodes (two records each, the default) are both conservative
    culling structures over the same triangle test: same hits as the oracle."""
    cam = default_camera(oracle, 16 / 9)
    o, d = raygen.camera_rays(cam, 1920, 1080, 150_000, seed=31)
    tmax = np.full(len(o), FLT_MAX, np.float32)
    c = cpu_matte.trace_closest(o, d, tmax)
    o2, d2 = raygen.bounce_rays(o, d, c[1], c[0] >= 0, seed=32)
    for wide in ("0", "1"):
        monkeypatch.setenv("RT_BVH_WIDE", wide)
        sc = api.Scene(bunny_matte)
        assert sc.info()["pairs"] > 1000
        _closest_compare(sc, cpu_matte, o, d, tmax)
        _closest_compare(sc, cpu_matte, o2, d2, np.full(len(o2), FLT_MAX, np.float32))
        sc.close()


def test_deep_bvh_four_bunnies(api, oracle):
    """BASELINE config 4's geometry (277 816 triangles, deeper tree): traversal parity and a small render."""
    from rtcuda_amd import scenes
    arrays = scenes.cornell_bunny("four_bunnies")
    gpu, cpu = api.Scene(arrays), oracle.scene(arrays)
    cam = default_camera(oracle, 16 / 9)
    o, d = raygen.camera_rays(cam, 1920, 1080, 200_000, seed=41)
    g, c, _ = _closest_compare(gpu, cpu, o, d, np.full(len(o), FLT_MAX, np.float32))
    o2, d2 = raygen.bounce_rays(o, d, c[1], c[0] >= 0, seed=42)
    _closest_compare(gpu, cpu, o2, d2, np.full(len(o2), FLT_MAX, np.float32))
    w, h, spp = 64, 36, 8
    img_c, _, st_c = cpu.render(cam, w, h, spp, threads=8)
    img_g, st_g = gpu.render(api.make_camera(aspect=w / h), w, h, spp)
    assert st_g["shade_events"] == st_c["sum_mat"] and st_g["any_rays"] == st_c["sum_ah"]
    assert _rms(img_g, img_c).max() < 2e-6


def _box_scene(lights_kind):
    """Bare Cornell box (12 triangles) with a chosen light set: edge cases of the light code paths."""
    from rtcuda_amd import scenes
    a = scenes.cornell_bunny("matte", bunny=False)
    if lights_kind == "point":      # Light::make_point_light (light.cuh:70-76): is_delta, no MIS block, no emission
        lights = np.zeros(1, scenes.LIGHT_DTYPE)
        lights[0] = (scenes.POINT_LIGHT, (0.7, 0.15, -0.6), -1, (0.5, 0.5, 0.5))
        a.lights = lights
        a.tri_light = np.full(a.n_tris, -1, np.int32)
    elif lights_kind == "mixed":    # one area light + one point light
        lights = np.zeros(2, scenes.LIGHT_DTYPE)
        lights[0] = a.lights[0]
        lights[1] = (scenes.POINT_LIGHT, (0.3, 0.8, -0.3), -1, (0.2, 0.3, 0.4))
        a.lights = lights
        a.tri_light = np.where(a.tri_light == 0, 0, -1).astype(np.int32)
    elif lights_kind == "none":     # num_lights == 0: mat() returns after the path ray (render.cuh:174-177)
        a.lights = np.zeros(0, scenes.LIGHT_DTYPE)
        a.tri_light = np.full(a.n_tris, -1, np.int32)
    return a


@pytest.mark.parametrize("kind", ["point", "mixed", "none"])
def test_light_code_paths_match_oracle(api, oracle, kind):
    arrays = _box_scene(kind)
    w, h, spp = 48, 48, 8
    img_c, _, st_c = oracle.scene(arrays).render(default_camera(oracle, 1.0), w, h, spp, threads=8)
    img_g, st_g = api.Scene(arrays).render(api.make_camera(aspect=1.0), w, h, spp)
    assert st_g["shade_events"] == st_c["sum_mat"] and st_g["any_rays"] == st_c["sum_ah"]
    assert st_g["emission_adds"] == st_c["emission_adds"] and st_g["shadow_adds"] == st_c["ah_adds"]
    assert _rms(img_g, img_c).max() < 2e-6
    if kind == "none":
        assert st_g["any_rays"] == 0 and not img_g.any()
    else:
        assert img_g.any()


def test_empty_and_single_triangle_scenes(api, oracle):
    from rtcuda_amd import scenes
    base = scenes.cornell_bunny("matte", bunny=False)
    empty = scenes.SceneArrays(tris=np.zeros((0, 9), np.float32), tri_material=np.zeros(0, np.int32),
                               tri_light=np.zeros(0, np.int32), materials=base.materials,
                               lights=np.zeros(0, scenes.LIGHT_DTYPE))
    img, st = api.Scene(empty).render(api.make_camera(aspect=1.0), 16, 16, 4)
    assert not img.any() and st["camera_rays"] == 16 * 16 * 4 and st["shade_events"] == 0
    one = scenes.SceneArrays(tris=base.tris[8:9].copy(), tri_material=np.array([2], np.int32),
                             tri_light=np.array([-1], np.int32), materials=base.materials,
                             lights=np.zeros(0, scenes.LIGHT_DTYPE))
    g = api.Scene(one)
    o, d = raygen.camera_rays(default_camera(oracle, 1.0), 64, 64, 5000, seed=51)
    _closest_compare(g, oracle.scene(one), o, d, np.full(len(o), FLT_MAX, np.float32), 0.0)


def test_deterministic_accumulation_is_bit_reproducible_and_shard_exact(api, oracle, gpu_full, cpu_full):
    """RT_FLAG_DETERMINISTIC / rt_render_shard_fixed: 64-bit fixed-point sums instead of float atomics.
    Same bits every run; the shards of a multi-GPU render add up to EXACTLY the single-GPU sums."""
    import torch
    w, h, spp = 96, 54, 8
    cam = api.make_camera(aspect=w / h)
    a, st_a = gpu_full.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    b, _ = gpu_full.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ref, _, st_c = cpu_full.render(default_camera(oracle, w / h), w, h, spp, threads=8)
    assert st_a["shade_events"] == st_c["sum_mat"]
    assert _rms(a, ref).max() < 2e-6
    full = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    gpu_full.render_shard_fixed(cam, w, h, spp, 0, 1, full.data_ptr())
    for shards in (2, 8):
        acc = torch.zeros_like(full)
        for r in range(shards):
            gpu_full.render_shard_fixed(cam, w, h, spp, r, shards, acc.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(acc, full)
    out = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    api.post_process_fixed(full.data_ptr(), out.data_ptr(), w * h, spp)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(h, w, 3).view(np.uint32), a.view(np.uint32))


def test_device_lbvh_builder_gives_the_same_image(api, oracle, cpu_matte, bunny_matte, monkeypatch):
    """the optional GPU BVH build (RT_BVH_BUILDER=lbvh).  A different tree, the same hits -- as the binary
    tree the device emits (RT_BVH_WIDE=0) and collapsed to the 4-wide format the kernels walk by default; the deep
    LBVH also drives the persistent kernel's traversal stack into its overflow column."""
    ref = api.Scene(bunny_matte)
    ref_info = ref.info()
    assert ref_info["builder"] == "sah"
    cam = default_camera(oracle, 16 / 9)
    o, d = raygen.camera_rays(cam, 1920, 1080, 150_000, seed=61)
    w2, h2, spp2 = 400, 300, 20  # 2.29 generations: all but the last in k_paths
    img_ref, st_ref = ref.render(api.make_camera(aspect=w2 / h2), w2, h2, spp2)
    for wide in ("1", "0"):
        monkeypatch.setenv("RT_BVH_BUILDER", "lbvh")
        monkeypatch.setenv("RT_BVH_WIDE", wide)
        sc = api.Scene(bunny_matte)
        monkeypatch.delenv("RT_BVH_BUILDER")
        monkeypatch.delenv("RT_BVH_WIDE")
        info = sc.info()
        assert info["builder"] == "lbvh" and info["leaves"] == bunny_matte.n_tris
        if wide == "0":
            assert info["pairs"] == bunny_matte.n_tris - 1
        else:  # two records per 4-wide node, fewer nodes than the binary tree has
            assert info["pairs"] % 2 == 0 and info["pairs"] // 2 < bunny_matte.n_tris - 1
        assert 0 < info["build_seconds"] < 0.5
        g, c, _ = _closest_compare(sc, cpu_matte, o, d, np.full(len(o), FLT_MAX, np.float32))
        o2, d2 = raygen.bounce_rays(o, d, c[1], c[0] >= 0, seed=62)
        _closest_compare(sc, cpu_matte, o2, d2, np.full(len(o2), FLT_MAX, np.float32))
        w, h, spp = 64, 36, 8
        img_c, _, st_c = cpu_matte.render(default_camera(oracle, w / h), w, h, spp, threads=8)
        img_g, st_g = sc.render(api.make_camera(aspect=w / h), w, h, spp)
        assert st_g["shade_events"] == st_c["sum_mat"] and st_g["any_rays"] == st_c["sum_ah"]
        assert _rms(img_g, img_c).max() < 2e-6
        img_m, st_m = sc.render(api.make_camera(aspect=w2 / h2), w2, h2, spp2)
        for k in ("camera_rays", "shade_events", "closest_rays", "any_rays", "emission_adds", "shadow_adds", "rr_draws"):
            assert st_m[k] == st_ref[k], (wide, k)
        assert _rms(img_m, img_ref).max() < 2e-6
        print("LBVH build", info["build_seconds"], "s (wide=%s); host SAH build" % wide, ref_info["build_seconds"], "s")
        sc.close()


def test_split_probe_counts_what_the_frame_traces(api, gpu_full, bunny_full_bsdf):
    """rt_split_probe (the stage split priced: tools/split_probe.py; a LAB entry point: include/rtcuda_amd_tools.h,
    librtcuda_amd_tools.so) on a small frame: the rays and shading records it dumps are the frame's own -- as many closest-hit
    rays as the round pipeline traces in those rounds, every shading record in one of the three material buckets -- and every
    timed stage reports a positive time."""
    w, h, spp = 320, 180, 64  # 3.5 generations
    lab_scene = api.Scene(bunny_full_bsdf, library=api.tools_lib())  # (the lab library works on scenes of its own)
    r = api.split_probe(lab_scene, api.make_camera(aspect=w / h), w, h, spp, target_rays=3_000_000)
    lab_scene.close()
    with pytest.raises(api.RtError):
        api.split_probe(gpu_full, api.make_camera(aspect=w / h), w, h, spp, target_rays=1000)  # a product-library scene
    assert r["rounds"] >= 2 and r["closest_rays"] + r["any_rays"] >= 3_000_000
    assert r["closest_rays"] <= r["rounds"] * api.W and 0 < r["any_rays"] < r["closest_rays"]
    shades = r["shades_matte"] + r["shades_mirror"] + r["shades_glass"]
    assert r["shades_matte"] > r["shades_mirror"] > 0 and r["shades_glass"] > 0
    assert r["any_rays"] <= shades <= r["closest_rays"]  # a shadow ray needs a shade; a shade needs a path ray that hit
    for k in ("s_advance_round0", "s_advance", "s_trace_pool", "s_shade_matte", "s_shade_mirror", "s_shade_glass"):
        assert r[k] > 0, k
    for wv in (8, 6, 5, 4):
        assert r[f"s_trace_closest_w{wv}"] > 0 and r[f"s_trace_any_w{wv}"] > 0 and r[f"trace_blocks_per_cu_w{wv}"] >= 1
    # the frame itself is untouched by the probe (its own context, its own buffers)
    img, st = gpu_full.render(api.make_camera(aspect=w / h), w, h, spp)
    assert st["camera_rays"] == w * h * spp


def test_rcp_exact_normal_is_the_ieee_quotient_on_this_chip(tmp_path):
    """ref_visible (rtcuda_amd.hip) needs 1 / d exactly as the reference's IEEE division gives it and computes it as
    v_rcp_f32 + one FMA Newton step (rt_device.h: rcp_exact_normal).  v_rcp_f32 is a hardware approximation, so the proof is
    exhaustive and runs here: every normal fp32 bit pattern with |x| < 2^126 (4.2 * 10^9 operands), compiled with the
    product's flags, against the compiler's `1.f / x`."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "rcp_exact_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off",
                           "-fno-fast-math", os.path.join(ROOT, "tests", "cpp", "rcp_exact_check.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    one_step, two_steps, patterns = (int(x) for x in out.stdout.split()[:3])
    assert out.returncode == 0 and one_step == 0 and two_steps == 0, out.stdout
    assert patterns > 4_000_000_000


def test_knobs_are_ignored_without_the_gate(api, gpu_full, monkeypatch):
    """A drop-in library must not change behaviour because some RT_* variable is set in the user's shell: the experiment knobs
    are read only under RTCUDA_EXPERIMENTAL=1.  RT_PERSISTENT=0 (one launch per round instead of the persistent kernel) shows
    in the statistics -- many more stage rounds -- exactly when the gate is on."""
    w, h, spp = 320, 180, 64  # 3.5 generations
    cam = api.make_camera(aspect=w / h)
    _, st_default = gpu_full.render(cam, w, h, spp)
    monkeypatch.setenv("RT_PERSISTENT", "0")
    monkeypatch.setenv("RT_MAJORITY", "0")
    _, st_gated = gpu_full.render(cam, w, h, spp)
    assert st_gated["iterations"] > 3 * st_default["iterations"]
    monkeypatch.delenv("RTCUDA_EXPERIMENTAL")
    _, st_ungated = gpu_full.render(cam, w, h, spp)
    assert st_ungated["iterations"] == st_default["iterations"]
    monkeypatch.setenv("RTCUDA_EXPERIMENTAL", "yes")  # (only the exact value 1 opens the gate)
    _, st_other = gpu_full.render(cam, w, h, spp)
    assert st_other["iterations"] == st_default["iterations"]
    for k in ("shade_events", "any_rays", "shadow_adds", "rr_draws"):
        assert st_default[k] == st_gated[k] == st_ungated[k]
