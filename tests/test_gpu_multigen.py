"""GPU parity of the PERSISTENT kernel (k_paths) against the CPU oracle.  Run with -m gpu.

A render of w*h*spp <= W = 1 048 576 camera rays has one generation only; that generation is the
final one and runs on the lockstep round pipeline (k_advance + k_trace), so k_paths parks every slot
at once.  Every case here has MORE than W camera rays -- 2 to 4.7 generations -- so that all but the
last generation run inside k_paths: its ADV / GEN / ANY / CLOSEST blocks, the pixel stepping of
gen() for generation >= 1 (add-and-carry, W / spp, 64-bit divide), the non-lockstep Russian-roulette
re-roll chain, the idle skip, the GEN / ADV routing at max_bounces, and the small-shard build
(MIN_WAVES = 2) that every rank of an 8-GPU run launches.  Reference lines being matched:
render.cuh:84-137 (init), :139-248 (mat), :250-275 (gen), :278-328 (ah / ch), :428-449 (host loop).

The oracle runs in its LITERAL mode here -- the reference's own tree, its fp32 slab test on exact boxes and its tree-order
tie rule (bvh.cuh:221-357, aabb_intersector.cuh:14-36, triangle.cuh:49) -- against the DEFAULT kernels, which make the
reference's decisions on their own walk (ref_visible in rtcuda_amd.hip).  RT_FLAG_WATERTIGHT (the triangle-list definition)
is held against the oracle's watertight mode by the tests that name it.

Bar: integer event totals EQUAL to the oracle's, image RMS < 2e-6 per channel (only the order of
the float atomics differs; north-star tolerance 1e-4), fixed-point sums bit-equal to the oracle's and between shardings.
"""
import os

import numpy as np
import pytest

from conftest import default_camera, oracle_render, oracle_scene

pytestmark = pytest.mark.gpu

W = 1 << 20


@pytest.fixture(scope="module")
def api():
    from rtcuda_amd import api as _api
    _api.lib()  # raises if the HIP library is missing: there is no fallback
    return _api


_scene_cache = {}


def _scenes(api, oracle, variant):
    """(GPU scene, oracle scene) of a BASELINE scene variant, built once per module."""
    if variant not in _scene_cache:
        from rtcuda_amd import scenes
        _scene_cache[variant] = api.Scene(scenes.cornell_bunny(variant))
    # Oracle in its LITERAL mode: about one path ray in 10^7 is decided differently by the reference's own BVH walk than by
    # exhaustive search over all triangles (its fp32 slab test on exact boxes drops a triangle the triangle test accepts),
    # and exact ties go to the triangle its walk tests last.  The default kernels reproduce both (tests/test_traversal_audit.py
    # replays every ray of literal renders through their CPU twin); the comparison here is strict.
    return _scene_cache[variant], oracle_scene(oracle, variant, False)


def _rms(a, b):
    """Per-channel RMS difference.  The reference's estimator itself yields a NaN contribution now and then (first
    seen: full_bsdf 300x200x48, pixel (184, 12) -- a glass path; render.cuh has no guard and SURVEY Appendix A.4
    names one such source); a NaN pixel must be a NaN pixel on BOTH sides and is left out of the RMS."""
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), (np.argwhere(na != nb)[:8], int(na.sum()), int(nb.sum()))
    assert na.sum() <= max(3, 1e-5 * a.size)
    d = np.where(na, 0.0, a.astype(np.float64) - np.where(nb, 0.0, b.astype(np.float64)))
    return np.sqrt(np.mean(d ** 2, axis=(0, 1)))


def _max_abs(a, b):
    """Largest per-channel difference over ALL pixels (NaN pixels, equal on both sides by _rms, count as 0).  Both sides
    decide ties the same way (the reference's: the default kernels re-trace such a ray through the reference's own tree), so
    no pixel is exempt."""
    return np.nan_to_num(np.abs(a.astype(np.float64) - b.astype(np.float64))).max()


def _assert_same_events(st_g, st_c, n_rays):
    assert st_g["camera_rays"] == n_rays
    assert st_g["shade_events"] == st_c["sum_mat"]
    assert st_g["any_rays"] == st_c["sum_ah"]
    assert st_g["emission_adds"] == st_c["emission_adds"]
    assert st_g["shadow_adds"] == st_c["ah_adds"]
    assert st_g["rr_draws"] == st_c["rr_draws"]
    assert st_c["ch_adds"] == 0  # the one ray kind the product does not trace never contributes (Appendix A.3)


def _generations(w, h, spp):
    return -(-(w * h * spp) // W)


MULTIGEN_CASES = [
    # variant, w, h, spp, max_bounces, seed
    ("full_bsdf", 480, 270, 32, 10, 1),       # 3.96 generations; spp | W and 480 | W/spp ... dpx = 128: carries
    ("full_bsdf", 300, 200, 48, 10, 1),       # spp does not divide W: 64-bit divide path at generation >= 1
    ("full_bsdf", 257, 130, 64, 10, 1),       # odd width, spp | W: add-and-carry pixel stepping (dpx = 193)
    ("full_bsdf", 512, 512, 9, 0, 1),         # no bounce at all: every closest hit routes straight to GEN / emission
    ("full_bsdf", 512, 512, 9, 1, 1),
    ("full_bsdf", 400, 300, 20, 20, 1),       # long Russian-roulette chains in the non-lockstep re-roll loop
    ("full_bsdf", 480, 270, 17, 10, 12345),   # another seed, odd spp
    ("matte", 256, 256, 40, 10, 1),           # C1's scene
    ("four_bunnies", 480, 270, 20, 10, 1),    # C4's geometry (deep BVH: global stack overflow in k_paths)
    ("sixteen_lights", 480, 270, 20, 10, 1),  # C5's lights
]


@pytest.mark.parametrize("variant,w,h,spp,max_bounces,seed", MULTIGEN_CASES)
def test_persistent_kernel_matches_oracle(api, oracle, variant, w, h, spp, max_bounces, seed):
    assert w * h * spp > W, "these cases must run k_paths for real"
    gpu, _ = _scenes(api, oracle, variant)
    img_c, _, st_c = oracle_render(oracle, variant, w, h, spp, max_bounces=max_bounces, seed=seed)
    img_g, st_g = gpu.render(api.make_camera(aspect=w / h), w, h, spp, max_bounces=max_bounces, seed=seed)
    _assert_same_events(st_g, st_c, w * h * spp)
    # the product's schedule: one persistent launch + the lockstep rounds of the final generation only
    assert st_g["iterations"] <= max_bounces + 2
    assert st_c["iterations"] > (max_bounces + 1) * (_generations(w, h, spp) - 1)
    rms = _rms(img_g, img_c)
    assert rms.max() < 2e-6, rms
    assert _max_abs(img_g, img_c) < 1e-4


WATERTIGHT_CASES = [("full_bsdf", 480, 270, 32, 10, 1), ("matte", 256, 256, 40, 10, 1), ("sixteen_lights", 480, 270, 20, 10, 1)]


@pytest.mark.parametrize("variant,w,h,spp,max_bounces,seed", WATERTIGHT_CASES)
def test_watertight_flag_matches_the_watertight_oracle(api, oracle, variant, w, h, spp, max_bounces, seed):
    """RT_FLAG_WATERTIGHT: no accepted hit is lost to a box test and ties go to the larger caller index -- what exhaustive
    search over the triangle list returns, and what the oracle's watertight mode restates.  Equal events, same image."""
    gpu, _ = _scenes(api, oracle, variant)
    img_c, _, st_c = oracle_render(oracle, variant, w, h, spp, max_bounces=max_bounces, seed=seed, watertight=True)
    img_g, st_g = gpu.render(api.make_camera(aspect=w / h), w, h, spp, max_bounces=max_bounces, seed=seed, flags=api.FLAG_WATERTIGHT)
    _assert_same_events(st_g, st_c, w * h * spp)
    assert st_g["literal_retraces"] == 0 and st_g["reference_lost_hits"] == 0  # (that machinery is not in this build)
    assert _rms(img_g, img_c).max() < 2e-6 and _max_abs(img_g, img_c) < 1e-4


def test_watertight_flag_differs_from_the_reference_only_by_audited_rays(api, oracle):
    """RT_FLAG_WATERTIGHT against the LITERAL oracle (the reference's own slab test): the event totals may differ by the
    few rays the reference's BVH walk loses (1 of 11.5 M path rays on this frame: tests/test_traversal_audit.py), nothing
    more -- and the default kernels' counters say which rays those were."""
    w, h, spp = 256, 256, 40
    gpu, _ = _scenes(api, oracle, "matte")
    img_c, _, st_c = oracle_render(oracle, "matte", w, h, spp, watertight=False)
    _, st_d = gpu.render(api.make_camera(aspect=1.0), w, h, spp)
    _assert_same_events(st_d, st_c, w * h * spp)
    assert 1 <= st_d["reference_lost_hits"] <= 4 and 1 <= st_d["literal_retraces"] <= 16, st_d
    img_g, st_g = gpu.render(api.make_camera(aspect=1.0), w, h, spp, flags=api.FLAG_WATERTIGHT)
    for kg, kc in (("shade_events", "sum_mat"), ("any_rays", "sum_ah"), ("emission_adds", "emission_adds"),
                   ("shadow_adds", "ah_adds"), ("rr_draws", "rr_draws")):
        assert abs(st_g[kg] - st_c[kc]) <= 4, (kg, st_g[kg], st_c[kc])
    d = np.abs(img_g.astype(np.float64) - img_c)
    assert (d.max(axis=2) > 1e-4).sum() <= 2  # pixels of the one diverging path
    assert np.sqrt(np.mean(d ** 2)) < 1e-4     # the north-star tolerance holds against the literal reference too


def test_persistent_kernel_deterministic_mode_matches_oracle(api, oracle):
    """RT_FLAG_DETERMINISTIC over 2.5 generations: bit-reproducible, and equal to the oracle within the
    fixed-point quantum."""
    w, h, spp = 256, 256, 40
    gpu, _ = _scenes(api, oracle, "matte")
    cam = api.make_camera(aspect=1.0)
    a, st_a = gpu.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    b, st_b = gpu.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    img_c, _, st_c = oracle_render(oracle, "matte", w, h, spp)
    _assert_same_events(st_a, st_c, w * h * spp)
    assert _rms(a, img_c).max() < 2e-6


def test_small_shard_build_of_the_persistent_kernel_matches_oracle(api, oracle):
    """8 slot-range shards of a 3.96-generation frame: each shard is what one rank of an 8-GPU run renders, and
    launches the MIN_WAVES = 2 build of k_paths (gen() inside ADV, LDS top of the tree).  Per-shard event totals
    and raw sums against the oracle's render of the same slot range; the shards' fixed-point sums add up to
    EXACTLY the unsharded frame."""
    import torch
    w, h, spp = 480, 270, 32
    shards = 8
    gpu, _ = _scenes(api, oracle, "full_bsdf")
    cam_g = api.make_camera(aspect=w / h)
    full = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    st_full = gpu.render_shard_fixed(cam_g, w, h, spp, 0, 1, full.data_ptr())
    acc = torch.zeros_like(full)
    tot = {k: 0 for k in ("camera_rays", "shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws")}
    n = W // shards
    for r in range(shards):
        part = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
        st = gpu.render_shard(cam_g, w, h, spp, r, shards, part.data_ptr())
        gpu.render_shard_fixed(cam_g, w, h, spp, r, shards, acc.data_ptr())
        for k in tot:
            tot[k] += st[k]
        if r in (0, 5):  # the oracle on the same slot range (two shards keep the CPU time down)
            _, raw_c, st_c = oracle_render(oracle, "full_bsdf", w, h, spp, slot_lo=r * n, slot_hi=(r + 1) * n)
            # (the oracle's shard run stops when ITS slots stop shading; the product's lockstep rounds do the same)
            assert st["shade_events"] == st_c["sum_mat"] and st["any_rays"] == st_c["sum_ah"]
            assert st["shadow_adds"] == st_c["ah_adds"] and st["rr_draws"] == st_c["rr_draws"]
            got = part.cpu().numpy().reshape(h, w, 3)
            assert np.allclose(got, raw_c, rtol=2e-5, atol=1e-6, equal_nan=True)
    torch.cuda.synchronize()
    for k in tot:
        assert tot[k] == st_full[k], k
    assert torch.equal(acc, full)
    img_c, _, st_c = oracle_render(oracle, "full_bsdf", w, h, spp)
    assert tot["shade_events"] == st_c["sum_mat"] and tot["any_rays"] == st_c["sum_ah"]
    out = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    api.post_process_fixed(acc.data_ptr(), out.data_ptr(), w * h, spp)
    torch.cuda.synchronize()
    assert _rms(out.cpu().numpy().reshape(h, w, 3), img_c).max() < 2e-6


def test_half_populated_waves_knob_gives_the_same_shard(api, oracle, monkeypatch):
    """RT_HALF_WAVES=1 (an experiment knob, measured slower at 1/8 and kept: profiles/r04_experiments.md): 32 slots per wave
    and twice the workgroups on shards of <= 1/8 of the slots -- the full-pool build of k_paths with idle upper half-waves.
    Same slots, same chains: the shard's event totals and fixed-point sums must not move."""
    import torch
    w, h, spp = 480, 270, 32
    gpu, _ = _scenes(api, oracle, "full_bsdf")
    cam = api.make_camera(aspect=w / h)
    keys = ("camera_rays", "shade_events", "closest_rays", "any_rays", "emission_adds", "shadow_adds", "rr_draws")
    for shards, r in ((8, 3), (16, 9)):
        a = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
        st_a = gpu.render_shard_fixed(cam, w, h, spp, r, shards, a.data_ptr())
        monkeypatch.setenv("RT_HALF_WAVES", "1")
        b = torch.zeros_like(a)
        st_b = gpu.render_shard_fixed(cam, w, h, spp, r, shards, b.data_ptr())
        monkeypatch.delenv("RT_HALF_WAVES")
        torch.cuda.synchronize()
        assert all(st_a[k] == st_b[k] for k in keys), (shards, st_a, st_b)
        assert torch.equal(a, b), shards


def test_lds_top_of_tree_on_small_shards(api, oracle, monkeypatch):
    """BVH nodes staged through LDS (north star; `top` / `top_n` of inner_step): the small-shard build of k_paths copies
    the first records of the tree -- its top levels in breadth-first order -- into LDS and walks them from there.  On by
    default for the 2-wide tree on a 1/8 shard (384 records), on request (RT_TOP_NODES) for the 4-wide tree.  Both are
    driven here and held against the oracle's render of the same slot range: equal event totals, same raw sums."""
    import torch
    from rtcuda_amd import scenes
    w, h, spp = 480, 270, 32
    shards = 8
    n = W // shards
    cam_g = api.make_camera(aspect=w / h)

    def check(gpu, r, expect_top):
        part = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
        st = gpu.render_shard(cam_g, w, h, spp, r, shards, part.data_ptr())
        torch.cuda.synchronize()
        assert st["lds_top_records"] == expect_top, st["lds_top_records"]
        _, raw_c, st_c = oracle_render(oracle, "full_bsdf", w, h, spp, slot_lo=r * n, slot_hi=(r + 1) * n)
        assert st["shade_events"] == st_c["sum_mat"] and st["any_rays"] == st_c["sum_ah"]
        assert st["emission_adds"] == st_c["emission_adds"]
        assert st["shadow_adds"] == st_c["ah_adds"] and st["rr_draws"] == st_c["rr_draws"]
        assert np.allclose(part.cpu().numpy().reshape(h, w, 3), raw_c, rtol=2e-5, atol=1e-6, equal_nan=True)

    gpu_wide, _ = _scenes(api, oracle, "full_bsdf")
    check(gpu_wide, 0, 0)                      # default: the 4-wide tree keeps its top in L2 (measured: no gain)
    monkeypatch.setenv("RT_TOP_NODES", "128")  # 64 four-wide nodes in LDS
    check(gpu_wide, 0, 128)
    check(gpu_wide, 5, 128)
    monkeypatch.delenv("RT_TOP_NODES")
    monkeypatch.setenv("RT_BVH_WIDE", "0")     # (read at scene creation)
    gpu_pairs = api.Scene(scenes.cornell_bunny("full_bsdf"))
    monkeypatch.delenv("RT_BVH_WIDE")
    check(gpu_pairs, 0, 384)
    check(gpu_pairs, 5, 384)
    monkeypatch.setenv("RT_STACK_CAP", "2")    # LDS top of the tree together with the global overflow stack
    check(gpu_pairs, 5, 384)
    monkeypatch.delenv("RT_STACK_CAP")
    gpu_pairs.close()


def test_deterministic_mode_on_a_frame_with_a_nan_contribution(api, oracle):
    """full_bsdf 300x200x48 has one pixel whose estimate is NaN in the reference's estimator (a glass path; see _rms).
    Default mode: the NaN reaches the framebuffer, as in the reference (three float atomicAdds, vec3.cuh:149-153).
    RT_FLAG_DETERMINISTIC (64-bit fixed-point sums) cannot hold a NaN and DROPS non-finite contributions -- documented at
    deposit() -- so there, and only there, the two modes differ: that pixel is finite, every other pixel is the same
    and the event totals are equal."""
    w, h, spp = 300, 200, 48
    gpu, _ = _scenes(api, oracle, "full_bsdf")
    cam = api.make_camera(aspect=w / h)
    img_c, _, st_c = oracle_render(oracle, "full_bsdf", w, h, spp)
    nan_c = np.isnan(img_c).any(axis=2)
    assert 1 <= nan_c.sum() <= 3
    img_d, st_d = gpu.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    img_f, st_f = gpu.render(cam, w, h, spp)
    _assert_same_events(st_d, st_c, w * h * spp)
    _assert_same_events(st_f, st_c, w * h * spp)
    assert np.array_equal(np.isnan(img_f).any(axis=2), nan_c)  # default mode: NaN where the oracle has NaN
    assert np.isfinite(img_d).all()                            # deterministic mode: the NaN contribution was dropped
    ok = ~nan_c
    d = img_d[ok].astype(np.float64) - img_c[ok]
    assert np.sqrt(np.mean(d ** 2)) < 2e-6 and np.abs(d).max() < 1e-4
    img_d2, _ = gpu.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    assert np.array_equal(img_d.view(np.uint32), img_d2.view(np.uint32))


def test_two_and_four_shards_sum_exactly(api, oracle):
    import torch
    w, h, spp = 300, 200, 48
    gpu, _ = _scenes(api, oracle, "full_bsdf")
    cam = api.make_camera(aspect=w / h)
    full = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    gpu.render_shard_fixed(cam, w, h, spp, 0, 1, full.data_ptr())
    for shards in (2, 4):
        acc = torch.zeros_like(full)
        for r in range(shards):
            gpu.render_shard_fixed(cam, w, h, spp, r, shards, acc.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(acc, full)


def test_round_pipeline_and_scheduling_variants_agree_over_generations(api, oracle, monkeypatch):
    """The frame as one persistent launch (default), as one launch per round (RT_PERSISTENT=0), and the scheduling
    knobs of k_paths are the same estimator over several generations: equal event totals, same image."""
    w, h, spp = 400, 300, 20  # 2.29 generations
    gpu, _ = _scenes(api, oracle, "full_bsdf")
    cam = api.make_camera(aspect=w / h)
    img_p, st_p = gpu.render(cam, w, h, spp)
    keys = ("camera_rays", "shade_events", "closest_rays", "any_rays", "emission_adds", "shadow_adds", "rr_draws")
    monkeypatch.setenv("RT_PERSISTENT", "0")
    img_r, st_r = gpu.render(cam, w, h, spp)
    monkeypatch.delenv("RT_PERSISTENT")
    for k in keys:
        assert st_p[k] == st_r[k], k
    assert st_r["iterations"] > st_p["iterations"]
    assert _rms(img_p, img_r).max() < 2e-6
    # material-sorted shading (k_advance<.., SORT>: a workgroup's slots partitioned by material kind with ballot + mbcnt;
    # RT_SORT_SHADE=1, measured slower and off by default) against slot order: which thread serves a slot changes nothing
    import torch
    det = {}
    for sort in ("1", "0"):
        monkeypatch.setenv("RT_PERSISTENT", "0")
        monkeypatch.setenv("RT_SORT_SHADE", sort)
        buf = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
        st_s = gpu.render_shard_fixed(cam, w, h, spp, 0, 1, buf.data_ptr())
        torch.cuda.synchronize()
        monkeypatch.delenv("RT_PERSISTENT")
        monkeypatch.delenv("RT_SORT_SHADE")
        for k in keys:
            assert st_s[k] == st_p[k], (sort, k)
        det[sort] = buf
    assert torch.equal(det["1"], det["0"])
    for env in ({"RT_MAJORITY": "0"}, {"RT_ADV_BATCH": "7", "RT_GEN_BATCH": "1"}, {"RT_PRIO_ROTATE": "0"},
                {"RT_PATHS_BLOCKS": "256"}, {"RT_SPLIT": "2"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        img_m, st_m = gpu.render(cam, w, h, spp)
        for k in env:
            monkeypatch.delenv(k)
        for k in keys:
            assert st_m[k] == st_p[k], (env, k)
        assert _rms(img_m, img_p).max() < 2e-6, env


def test_concurrent_sub_shards_keep_their_own_overflow_stacks(api, oracle, monkeypatch):
    """RT_SPLIT=2 runs two k_paths grids at once on one device.  The deep four-bunny tree overflows the 8 LDS
    stack entries of k_paths routinely, so the grids must not share overflow columns."""
    w, h, spp = 480, 270, 20
    gpu, _ = _scenes(api, oracle, "four_bunnies")
    cam = api.make_camera(aspect=w / h)
    img_1, st_1 = gpu.render(cam, w, h, spp)
    monkeypatch.setenv("RT_SPLIT", "2")
    img_2, st_2 = gpu.render(cam, w, h, spp)
    monkeypatch.delenv("RT_SPLIT")
    for k in ("camera_rays", "shade_events", "closest_rays", "any_rays", "emission_adds", "shadow_adds", "rr_draws"):
        assert st_1[k] == st_2[k], k
    assert _rms(img_1, img_2).max() < 2e-6


def test_overflow_stack_path_gives_the_same_frame(api, oracle, monkeypatch):
    """RT_STACK_CAP=2 leaves two traversal-stack entries per lane in LDS, so nearly every push and pop of the 4-wide walk
    (up to three pushes per node step, stored without a branch: push_if4) goes through the global overflow column --
    in the persistent kernel, in the lockstep rounds and in the stage-level trace.  Same events, same image."""
    w, h, spp = 400, 300, 20
    for variant in ("full_bsdf", "four_bunnies"):
        gpu, _ = _scenes(api, oracle, variant)
        cam = api.make_camera(aspect=w / h)
        img_1, st_1 = gpu.render(cam, w, h, spp)
        monkeypatch.setenv("RT_STACK_CAP", "2")
        img_2, st_2 = gpu.render(cam, w, h, spp)
        monkeypatch.setenv("RT_BVH_WIDE", "0")  # (a new scene: the binary tree through the same two entries)
        from rtcuda_amd import scenes
        gpu_b = api.Scene(scenes.cornell_bunny(variant))
        img_3, st_3 = gpu_b.render(cam, w, h, spp)
        gpu_b.close()
        monkeypatch.delenv("RT_BVH_WIDE")
        monkeypatch.delenv("RT_STACK_CAP")
        for k in ("camera_rays", "shade_events", "closest_rays", "any_rays", "emission_adds", "shadow_adds", "rr_draws"):
            assert st_1[k] == st_2[k] == st_3[k], (variant, k)
        assert _rms(img_1, img_2).max() < 2e-6 and _rms(img_1, img_3).max() < 2e-6


GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_goldens.npz"))


@pytest.mark.parametrize("variant,w,h,spp", [("full_bsdf", 128, 72, 256), ("matte", 100, 60, 400)])
def test_multi_generation_render_matches_committed_golden(api, variant, w, h, spp):
    """2.25- and 2.29-generation frames against the committed fixtures (no oracle at run time)."""
    from rtcuda_amd import scenes
    key = f"{variant}_{w}x{h}x{spp}"
    sc = api.Scene(scenes.cornell_bunny(variant))
    img, st = sc.render(api.make_camera(aspect=w / h), w, h, spp, flags=api.FLAG_WATERTIGHT)  # (fixtures of the watertight oracle)
    assert [st["shade_events"], st["any_rays"], st["emission_adds"], st["shadow_adds"], st["rr_draws"],
            st["camera_rays"]] == GOLDEN[key + "_counts"].tolist()
    ref = GOLDEN[key + "_img"].astype(np.float32)
    assert _rms(img, ref).max() < 2e-6
    assert _max_abs(img, ref) < 1e-4


@pytest.mark.parametrize("variant,w,h,spp", [("matte", 160, 100, 160), ("full_bsdf", 128, 72, 256)])
def test_multi_generation_render_against_the_literal_reference_fixture(api, variant, w, h, spp):
    """The committed LITERAL-oracle fixtures (the reference's own fp32 slab test and tree-order tie rule).  Round 5: the
    default kernels make the reference's decisions, so the counts are EQUAL and no pixel moves (until round 4 the bound was
    the audited one: 4 events, 2 pixels over 1e-4); RT_FLAG_WATERTIGHT stays inside that bound."""
    from rtcuda_amd import scenes
    key = f"literal_{variant}_{w}x{h}x{spp}"
    sc = api.Scene(scenes.cornell_bunny(variant))
    img, st = sc.render(api.make_camera(aspect=w / h), w, h, spp)
    img_w, st_w = sc.render(api.make_camera(aspect=w / h), w, h, spp, flags=api.FLAG_WATERTIGHT)
    sc.close()
    want = GOLDEN[key + "_counts"].tolist()
    ref = GOLDEN[key + "_img"]
    got = [st["shade_events"], st["any_rays"], st["emission_adds"], st["shadow_adds"], st["rr_draws"], st["camera_rays"]]
    assert got == want
    assert _rms(img, ref).max() < 2e-6 and _max_abs(img, ref) < 1e-4
    got = [st_w["shade_events"], st_w["any_rays"], st_w["emission_adds"], st_w["shadow_adds"], st_w["rr_draws"], st_w["camera_rays"]]
    assert got[5] == want[5]
    assert all(abs(g - c) <= 4 for g, c in zip(got, want)), (got, want)
    assert np.array_equal(np.isnan(img_w), np.isnan(ref))
    d = np.nan_to_num(np.abs(img_w.astype(np.float64) - ref))
    assert (d.max(axis=2) > 1e-4).sum() <= 2
    assert np.sqrt(np.mean(d ** 2)) < 1e-4


def test_per_sample_rng_mode_is_partition_invariant_and_statistically_equivalent(api):
    """RT_FLAG_RNG_PER_SAMPLE (NOT the reference's random numbers: SURVEY section 7 "per_sample"): every camera ray has a
    stream of its own, keyed by (seed, camera ray id).
      * partition invariance: rank r of R renders all pixels at spp / R with the full slot pool; the ranks' fixed-point
        sums add up to EXACTLY the 1-rank sums for R = 2, 4, 8, and so do the event totals;
      * it is the same estimator: against the default (reference) mode the image differs by Monte-Carlo noise only -- as
        much as two default-mode renders with different seeds differ -- and the mean radiance agrees."""
    import torch
    from rtcuda_amd import scenes
    w, h, spp = 240, 135, 256  # 7.9 generations
    gpu = api.Scene(scenes.cornell_bunny("full_bsdf"))
    cam = api.make_camera(aspect=w / h)
    F = api.FLAG_RNG_PER_SAMPLE
    keys = ("camera_rays", "shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws")
    full = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    st_full = gpu.render_shard_fixed(cam, w, h, spp, 0, 1, full.data_ptr(), flags=F)
    assert st_full["camera_rays"] == w * h * spp
    for shards in (2, 4, 8):
        acc = torch.zeros_like(full)
        tot = {k: 0 for k in keys}
        for r in range(shards):
            st = gpu.render_shard_fixed(cam, w, h, spp, r, shards, acc.data_ptr(), flags=F)
            assert st["camera_rays"] == w * h * spp // shards  # every rank: all pixels, spp / R samples each
            for k in keys:
                tot[k] += st[k]
        torch.cuda.synchronize()
        assert torch.equal(acc, full), shards
        assert all(tot[k] == st_full[k] for k in keys), (shards, tot, st_full)
    # spp must split evenly over the ranks
    with pytest.raises(api.RtError):
        gpu.render_shard_fixed(cam, w, h, 255, 0, 2, full.data_ptr(), flags=F)
    gpu.close()
    # the estimator: per-sample streams vs the reference's slot streams, on the matte scene (no specular fireflies) and with
    # a robust measure -- the median absolute pixel difference -- next to the difference of two reference-mode renders
    gpu = api.Scene(scenes.cornell_bunny("matte"))
    img_ps, st_ps = gpu.render(cam, w, h, spp, flags=F)
    img_a, st_a = gpu.render(cam, w, h, spp, seed=1)
    img_b, _ = gpu.render(cam, w, h, spp, seed=2)
    gpu.close()

    def mad(x, y):
        return float(np.nanmedian(np.abs(x.astype(np.float64) - y)))
    noise = mad(img_a, img_b)  # two independent reference-mode renders
    assert noise > 0
    assert 0.8 * noise < mad(img_ps, img_a) < 1.25 * noise, (mad(img_ps, img_a), noise)
    assert abs(np.nanmean(img_ps) - np.nanmean(img_a)) < 0.005 * np.nanmean(img_a)
    assert abs(st_ps["shade_events"] - st_a["shade_events"]) < 0.005 * st_a["shade_events"]  # same path statistics


FIXED_CASES = [
    # variant, w, h, spp, mode: "default" (the reference's decisions on the product's walk) and "reference_walk"
    # (RT_FLAG_REFERENCE_WALK: every ray through the reference's own tree) against the LITERAL oracle, "watertight"
    # (RT_FLAG_WATERTIGHT) against the watertight oracle
    ("full_bsdf", 300, 200, 48, "default"),      # 2.7 generations, spp does not divide W, one NaN contribution (dropped)
    ("full_bsdf", 160, 90, 16, "default"),       # one generation: every camera ray is of the FINAL generation (lockstep pipeline)
    ("matte", 256, 256, 40, "default"),          # (a frame on which the reference loses a hit)
    ("sixteen_lights", 480, 270, 20, "default"),
    ("sixteen_lights", 480, 270, 48, "default"),  # 7 shadow rays whose occluder the reference's walk cannot see
    ("four_bunnies", 480, 270, 20, "default"),   # the deep tree (global overflow stack in k_paths) and the reference's depth-22 tree
    ("matte", 256, 256, 40, "reference_walk"),
    ("sixteen_lights", 480, 270, 20, "reference_walk"),
    ("four_bunnies", 480, 270, 20, "reference_walk"),
    ("full_bsdf", 300, 200, 48, "watertight"),
    ("matte", 256, 256, 40, "watertight"),
    ("sixteen_lights", 480, 270, 20, "watertight"),
]


@pytest.mark.parametrize("variant,w,h,spp,mode", FIXED_CASES)
def test_deterministic_sums_equal_the_oracles_fixed_point_sums_bit_for_bit(api, oracle, variant, w, h, spp, mode):
    """The IMAGE, bit for bit: RT_FLAG_DETERMINISTIC's int64 sums (2^-30 fixed point; a camera ray's contributions summed in
    float in path order, then converted -- contribution by contribution in the final generation) against the oracle's
    restatement of that accumulation over ITS paths (oracle.cpp render_literal, `fb_fixed`).  Integer adds commute, so there
    is exactly one right answer per frame, and every one of the w * h * 3 sums must equal it: every path, every contribution,
    every rounding.  Default kernels and RT_FLAG_REFERENCE_WALK vs the literal oracle; RT_FLAG_WATERTIGHT vs the watertight
    oracle.  The six full BASELINE frames are held to committed hashes of the same arrays
    (test_every_full_baseline_frame_image_hash)."""
    import torch
    from conftest import usable_cpus
    from oracle.oracle import sums_hash
    gpu, _ = _scenes(api, oracle, variant)
    osc = oracle_scene(oracle, variant, mode == "watertight")
    flags = {"default": 0, "reference_walk": api.FLAG_REFERENCE_WALK, "watertight": api.FLAG_WATERTIGHT}[mode]
    want = np.zeros((h, w, 3), np.int64)
    _, _, st_c = osc.render(default_camera(oracle, w / h), w, h, spp, threads=usable_cpus(), fixed_out=want)
    got = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    st_g = gpu.render_shard_fixed(api.make_camera(aspect=w / h), w, h, spp, 0, 1, got.data_ptr(), flags=flags)
    torch.cuda.synchronize()
    _assert_same_events(st_g, st_c, w * h * spp)
    g = got.cpu().numpy().reshape(h, w, 3)
    assert np.array_equal(g, want), (int((g != want).sum()), np.argwhere(g != want)[:4])
    assert sums_hash(g) == sums_hash(want)
    # ... and the same array from 8 slot-range shards (what 8 ranks would reduce)
    acc = torch.zeros_like(got)
    for r in range(8):
        gpu.render_shard_fixed(api.make_camera(aspect=w / h), w, h, spp, r, 8, acc.data_ptr(), flags=flags)
    torch.cuda.synchronize()
    assert torch.equal(acc, got)


@pytest.mark.gpu
@pytest.mark.parametrize("lookfrom,vfov", [((0.5, 0.5, 40.0), 1.6), ((-30.0, 12.0, 25.0), 2.2)])
def test_a_camera_far_outside_the_scene_re_pads_the_node_records_and_changes_nothing(api, oracle, lookfrom, vfov):
    """The 4-wide node step's one-fma plane distance is exact enough only for ray origins the records are padded for
    (rt_bvh.h, pad_quads_for_origins): a camera ~40 scene sizes away makes rt_render_shard re-pad and upload them
    (ensure_origin_radius) -- on a scene object that renders from the usual camera before and after.  Every frame must be
    the literal oracle's, bit for bit, and RT_FLAG_REFERENCE_WALK's."""
    import torch
    from conftest import usable_cpus
    variant, w, h, spp = "full_bsdf", 320, 180, 24
    gpu, _ = _scenes(api, oracle, variant)
    osc = oracle_scene(oracle, variant, False)
    near = api.make_camera(aspect=w / h)

    def sums(cam, flags=0):
        buf = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
        st = gpu.render_shard_fixed(cam, w, h, spp, 0, 1, buf.data_ptr(), flags=flags)
        torch.cuda.synchronize()
        return buf.cpu().numpy().reshape(h, w, 3), st

    before, _ = sums(near)
    far = api.make_camera(lookfrom=lookfrom, lookat=(0.5, 0.4, -0.5), vfov=vfov, aspect=w / h)
    want = np.zeros((h, w, 3), np.int64)
    ocam = oracle.camera(lookfrom, (0.5, 0.4, -0.5), (0.0, 1.0, 0.0), vfov, w / h)
    _, _, st_c = osc.render(ocam, w, h, spp, threads=usable_cpus(), fixed_out=want)
    got, st_g = sums(far)
    _assert_same_events(st_g, st_c, w * h * spp)
    assert st_c["sum_mat"] > 0.2 * w * h * spp  # (the frame looks at the scene)
    assert np.array_equal(got, want), int((got != want).sum())
    ref, _ = sums(far, api.FLAG_REFERENCE_WALK)
    assert np.array_equal(ref, want)
    after, _ = sums(near)  # the wider padding stays: still the same image from the usual camera
    assert np.array_equal(after, before)


def _full_size_hashes():
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "golden", "full_size_image_hashes.json")
    return json.load(open(path))["frames"] if os.path.exists(path) else []


def _full_size_hash_cases():
    out = []
    for f in _full_size_hashes():
        out.append((f, "default" if f["mode"] == "literal" else "watertight"))
        if f["mode"] == "literal":
            out.append((f, "reference_walk"))
    return out


@pytest.mark.parametrize("frame,how", _full_size_hash_cases(),
                         ids=lambda x: x if isinstance(x, str) else f"{x['scene']}_{x['width']}x{x['height']}x{x['spp']}_{x['mode']}")
def test_every_full_baseline_frame_image_hash(api, frame, how):
    """Full-size image parity inside the GPU suite: the int64 fixed-point sums of a whole BASELINE frame (6 220 800 values)
    hashed to 64 bits and compared with the committed hash of the ORACLE's array for that frame (tests/golden/make_full_size_hashes.py:
    minutes of CPU per frame; tests/golden/full_size_image_hashes.json).  `mode` literal: the DEFAULT kernels -- the ones
    bench.py times -- and RT_FLAG_REFERENCE_WALK; `mode` watertight: RT_FLAG_WATERTIGHT.  Equal hashes = every pixel of the
    frame bit-equal to the oracle's."""
    import torch
    from oracle.oracle import sums_hash
    from rtcuda_amd import scenes
    w, h, spp = frame["width"], frame["height"], frame["spp"]
    if frame["scene"] not in _scene_cache:
        _scene_cache[frame["scene"]] = api.Scene(scenes.cornell_bunny(frame["scene"]))
    gpu = _scene_cache[frame["scene"]]
    got = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    flags = {"default": 0, "reference_walk": api.FLAG_REFERENCE_WALK, "watertight": api.FLAG_WATERTIGHT}[how]
    st = gpu.render_shard_fixed(api.make_camera(aspect=w / h), w, h, spp, 0, 1, got.data_ptr(), flags=flags)
    torch.cuda.synchronize()
    for k, v in frame["events"].items():
        assert st[k] == v, (k, st[k], v)
    assert sums_hash(got.cpu().numpy()) == frame["sums_sha256_64"]


def _full_size_frames():
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    return json.load(open(os.path.join(here, "golden", "full_size_event_totals.json")))["frames"]


@pytest.mark.parametrize("column", ["oracle_literal", "oracle_watertight"])
@pytest.mark.parametrize("frame", _full_size_frames(), ids=lambda f: f"{f['scene']}_{f['width']}x{f['height']}x{f['spp']}")
def test_every_full_baseline_frame_has_the_oracles_event_totals(api, frame, column):
    """The six FULL BASELINE frames (1920x1080 at 256 / 512 / 1024 spp: 506 - 2 025 generations, 2 - 8 * 10^9 rays), each
    rendered once by k_paths: the five integer event totals EQUAL the oracle's -- the default kernels the LITERAL oracle's,
    RT_FLAG_WATERTIGHT the watertight oracle's.  The oracle's totals are committed answers (tests/golden/full_size_event_totals.json, made
    by tools/full_size_parity.py + tests/golden/make_full_size_totals.py: 100 - 400 s of 16 cores per frame and mode).  The
    image of a frame of this size is held against the oracle in profiles/r03_full_size_parity*.json; here: the oracle's
    number of NaN pixels, nothing negative, and the totals -- a checksum over every scheduling decision, every
    random number and every ray of the frame."""
    import torch
    from rtcuda_amd import scenes
    w, h, spp = frame["width"], frame["height"], frame["spp"]
    if frame["scene"] not in _scene_cache:
        _scene_cache[frame["scene"]] = api.Scene(scenes.cornell_bunny(frame["scene"]))
    gpu = _scene_cache[frame["scene"]]
    fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    st = gpu.render_shard(api.make_camera(aspect=w / h), w, h, spp, 0, 1, fb.data_ptr(),
                          flags=0 if column == "oracle_literal" else api.FLAG_WATERTIGHT)
    torch.cuda.synchronize()
    assert st["camera_rays"] == frame["samples"] == w * h * spp
    for k, v in frame[column].items():
        assert st[k] == v, (k, st[k], v)
    # (RT_FLAG_REFERENCE_WALK holds the literal column too:
    # tests/test_gpu_reference_walk.py::test_every_full_baseline_frame_equals_the_literal_oracle_under_the_reference_walk)
    if column == "oracle_literal":
        assert st["literal_retraces"] < 1e-6 * (st["closest_rays"] + st["any_rays"])  # the rare path is rare
    # the reference's estimator yields a NaN contribution about once in 10^7 samples (render.cuh has no guard; SURVEY Appendix
    # A.4 names one source): as many NaN pixels as the oracle's frame has, and nothing negative
    nan_pixels = int(torch.isnan(fb).view(-1, 3).any(dim=1).sum().item())
    want_nan = {(f["scene"], f["spp"], f["mode"]): f["nan_pixels_float_image"] for f in _full_size_hashes()}
    want_nan = want_nan.get((frame["scene"], spp, "literal" if column == "oracle_literal" else "watertight"), frame["oracle_nan_pixels"])
    assert nan_pixels == want_nan, (nan_pixels, want_nan)
    assert bool((torch.nan_to_num(fb) >= 0).all().item())


@pytest.mark.parametrize("shards", [2, 8])
def test_full_baseline_frame_is_exactly_the_sum_of_its_slot_shards(api, shards):
    """BASELINE configs[1] at full size (1920x1080x256, full BSDF) in RT_FLAG_DETERMINISTIC arithmetic: the int64 raw sums of
    the R slot-range shards -- what the R ranks of a multi-GPU run render and sum-reduce -- add up to EXACTLY the unsharded
    frame's, and so do the event totals (which the test above holds equal to the oracle's).  R = 8 launches the small-shard
    build of k_paths (2 waves per SIMD, gen() inside the ADV block), R = 2 the full-occupancy build on half the slots."""
    import torch
    from rtcuda_amd import scenes
    w, h, spp = 1920, 1080, 256
    if "full_bsdf" not in _scene_cache:
        _scene_cache["full_bsdf"] = api.Scene(scenes.cornell_bunny("full_bsdf"))
    gpu = _scene_cache["full_bsdf"]
    cam = api.make_camera(aspect=w / h)
    full = torch.zeros(h * w * 3, dtype=torch.int64, device="cuda")
    st_full = gpu.render_shard_fixed(cam, w, h, spp, 0, 1, full.data_ptr())
    acc = torch.zeros_like(full)
    keys = ("camera_rays", "shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws")
    tot = {k: 0 for k in keys}
    for r in range(shards):
        st = gpu.render_shard_fixed(cam, w, h, spp, r, shards, acc.data_ptr())
        for k in keys:
            tot[k] += st[k]
    torch.cuda.synchronize()
    assert tot == {k: st_full[k] for k in keys}
    assert tot["camera_rays"] == w * h * spp
    assert torch.equal(acc, full)


@pytest.mark.parametrize("scene,spp", [("matte", 1024), ("sixteen_lights", 512), ("four_bunnies", 256)])
def test_eight_rank_shards_of_the_other_baseline_configs_add_up_to_the_oracles_totals(api, scene, spp):
    """What every rank of the driver's 8-GPU run renders for bench.py's `extra_configs` (BASELINE configs 3, 5, 4), one rank
    after the other on this GPU: the small-shard build of k_paths on each of the 8 slot ranges of the full frame.  The event
    totals summed over the shards equal the committed oracle totals of the frame -- the check bench.py itself makes on the
    node -- and the summed camera rays are the frame's."""
    import json
    import torch
    from rtcuda_amd import scenes
    here = os.path.dirname(os.path.abspath(__file__))
    want = [f for f in json.load(open(os.path.join(here, "golden", "full_size_event_totals.json")))["frames"]
            if (f["scene"], f["spp"]) == (scene, spp)][0]
    w, h = 1920, 1080
    if scene not in _scene_cache:
        _scene_cache[scene] = api.Scene(scenes.cornell_bunny(scene))
    gpu = _scene_cache[scene]
    cam = api.make_camera(aspect=w / h)
    fb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    keys = ("camera_rays", "shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws")
    tot = {k: 0 for k in keys}
    for r in range(8):
        st = gpu.render_shard(cam, w, h, spp, r, 8, fb.data_ptr())
        for k in keys:
            tot[k] += st[k]
    torch.cuda.synchronize()
    assert tot["camera_rays"] == w * h * spp
    for k, v in want["oracle_literal"].items():  # (the default kernels: the LITERAL oracle's totals)
        assert tot[k] == v, (k, tot[k], v)
    nan_pixels = int(torch.isnan(fb).view(-1, 3).any(dim=1).sum().item())
    want_nan = {(f["scene"], f["spp"], f["mode"]): f["nan_pixels_float_image"] for f in _full_size_hashes()}
    assert nan_pixels == want_nan.get((scene, spp, "literal"), want["oracle_nan_pixels"])
