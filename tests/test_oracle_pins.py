"""CPU tests: the oracle against every known answer SURVEY.md Appendix C records (tests/golden/appendix_c.json).

The reference ships no tests, fixtures or golden images, and it cannot be built in this image
(nvcc, cuRAND and CUB are absent), so these values -- produced by the surveyor from the reference's
own headers -- are the only pins the oracle has.  The image-level numbers were recorded with glibc's
sincosf/powf, so they are checked against the "libm" flavour of the oracle; the "pinned" flavour
(what the GPU is compared with bit for bit) differs from it only in those two functions.
"""
import json
import os

import numpy as np
import pytest

from conftest import default_camera, usable_cpus

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "appendix_c.json")) as fh:
    APPX = json.load(fh)


def _hex_u32(a):
    return ["%08x" % int(x) for x in np.asarray(a).view(np.uint32).ravel()]


def test_xorwow_known_answers(oracle):
    for sub, rec in APPX["xorwow_seed1"].items():
        st = oracle.xorwow_init(1, int(sub))
        if "state" in rec:
            assert _hex_u32(st) == rec["state"]
        raw, uni = oracle.xorwow_draw(st, 3)
        assert _hex_u32(raw) == rec["raw"]
        if "uniform" in rec:
            assert np.allclose(uni, rec["uniform"], rtol=0, atol=5e-10)
    for x, bits in APPX["uniform_edges"].items():
        u = np.float32(oracle.lib.orc_uniform_from_u32(int(x, 16)))
        assert _hex_u32(u) == [bits]


def test_xorwow_range_equals_single_init(oracle):
    rng = oracle.xorwow_init_range(1, 4090, 12)
    for k in range(12):
        assert np.array_equal(rng[k], oracle.xorwow_init(1, 4090 + k))
    # the jump map is linear over GF(2): J(a ^ b) == J(a) ^ J(b)
    rows = oracle.jump_rows()
    assert rows.shape == (160, 5) and rows.any()


def test_camera_known_answers(oracle):
    rec = APPX["camera_aspect_1"]
    cam = default_camera(oracle, 1.0)
    assert _hex_u32(cam[3:6]) == rec["upper_left"]
    assert _hex_u32(cam[6:7]) == [rec["horizontal_x"]]
    assert _hex_u32(cam[10:11]) == [rec["vertical_y"]]
    ray = oracle.camera_get_ray(cam, 0.25, 0.75)
    assert _hex_u32(ray[3:6]) == rec["get_ray_0.25_0.75_dir"]
    rec = APPX["camera_aspect_16_9"]
    cam = default_camera(oracle, 16.0 / 9.0)
    assert np.allclose(cam[3:6], rec["upper_left"], rtol=0, atol=5e-9)
    assert abs(cam[6] - rec["horizontal_x"]) < 5e-8 and abs(cam[10] - rec["vertical_y"]) < 5e-9


def test_triangle_known_answers(oracle):
    rec = APPX["light_triangle"]
    tri, area = oracle.triangle(rec["p"])
    assert np.allclose(tri[3:6], rec["e1"], rtol=0, atol=1e-9)
    assert np.allclose(tri[6:9], rec["e2"], rtol=0, atol=1e-9)
    assert np.allclose(tri[9:12], rec["n"], rtol=0, atol=1e-9)
    assert abs(area - rec["area"]) < 1e-9
    hit, tuv = oracle.triangle_intersect(rec["p"], rec["ray_o"], rec["ray_d"])
    assert hit
    assert np.allclose(tuv, [rec["t"], rec["u"], rec["v"]], rtol=0, atol=5e-9)
    # edge cases of the acceptance test (triangle.cuh:47-49): behind the origin, beyond tmax, parallel
    assert not oracle.triangle_intersect(rec["p"], rec["ray_o"], [0, -1, 0])[0]
    assert not oracle.triangle_intersect(rec["p"], rec["ray_o"], rec["ray_d"], tmax=0.5)[0]
    assert not oracle.triangle_intersect(rec["p"], rec["ray_o"], [1, 0, 0])[0]
    assert oracle.triangle_intersect(rec["p"], rec["ray_o"], rec["ray_d"], tmax=float(tuv[0]))[0]  # t <= tmax


def test_offset_ray_origin_and_power_heuristic(oracle):
    for rec in APPX["offset_ray_origin"]:
        assert _hex_u32(oracle.offset_ray_origin(rec["p"], rec["n"])) == rec["out"]
    for rec in APPX["power_heuristic"]:
        assert abs(oracle.power_heuristic(rec["f"], rec["g"]) - rec["out"]) < 5e-10
    # the int-truncating second argument (utility.cuh:53): any g in [0, 1) behaves as 0
    assert oracle.power_heuristic(0.7, 0.999) == 1.0


def test_bvh_builder_known_answers(oracle, bunny_matte):
    rec = APPX["bvh"]["matte"]
    st = oracle.scene(bunny_matte).bvh_stats()
    assert st["num_nodes"] == rec["num_nodes"] and st["num_prims"] == rec["num_prims"]
    assert st["max_depth"] == rec["max_depth"] and st["num_leaves"] == rec["num_leaves"]
    assert st["leaf_hist"][:5] == rec["leaf_hist"]
    assert np.array_equal(st["root_bounds"], np.array(rec["root_bounds"], np.float32))


@pytest.mark.parametrize("variant", ["four_bunnies", "sixteen_lights"])
def test_bvh_builder_variants(oracle, variant):
    from rtcuda_amd import scenes
    st = oracle.scene(scenes.cornell_bunny(variant)).bvh_stats()
    rec = APPX["bvh"][variant]
    assert (st["num_nodes"], st["num_prims"], st["max_depth"]) == (rec["num_nodes"], rec["num_prims"], rec["max_depth"])


@pytest.mark.parametrize("rec", [r for r in APPX["images_matte"] if r["w"] * r["h"] <= 65536],
                         ids=lambda r: f"{r['w']}x{r['h']}x{r['spp']}")
def test_image_goldens_matte(oracle_libm, bunny_matte, rec):
    sc = oracle_libm.scene(bunny_matte)
    w, h = rec["w"], rec["h"]
    img, _, st = sc.render(default_camera(oracle_libm, w / h), w, h, rec["spp"], threads=8)
    assert st["iterations"] == rec["iterations"]
    assert (st["sum_mat"], st["sum_gen"], st["sum_ah"], st["sum_ch"]) == (rec["mat"], rec["gen"], rec["ah"], rec["ch"])
    mean = img.reshape(-1, 3).mean(0, dtype=np.float64)
    assert np.allclose(mean, rec["mean"], rtol=0, atol=6e-10), mean
    if "iter_counts" in rec:
        assert np.array_equal(st["iter_counts"], np.array(rec["iter_counts"], np.int32))
        assert (st["emission_adds"], st["ah_adds"], st["ch_adds"]) == (rec["emission"], rec["ah_adds"], rec["ch_adds"])
    assert st["ch_adds"] == 0  # the BSDF-sampled MIS ray never contributes
    assert np.isfinite(img).all()


@pytest.mark.xfail(strict=False, reason="SURVEY Appendix C's FNV-1a-64 image hashes are not reproduced: the survey session's hash "
                                        "definition (or its libm) cannot be recovered offline; counts and mean RGB do match")
@pytest.mark.parametrize("rec", [r for r in APPX["images_matte"] if r["w"] * r["h"] <= 4096],
                         ids=lambda r: f"{r['w']}x{r['h']}x{r['spp']}")
def test_appendix_c_image_hashes(oracle_libm, bunny_matte, rec):
    """The image hashes of SURVEY Appendix C ("FNV-1a-64 over the fp32 words", libm-dependent) are the one family of
    known answers the oracle does NOT reproduce.  Tried: the libm flavour with one thread -- whose
    deposits are applied in exactly the serial launch order of the survey's shim (init() emission in slot order, then ah()
    and ch() in compacted queue order: render.cuh:98-103,291-293,321-326) -- hashed as bytes, as 32-bit words, big-endian,
    FNV-1 and FNV-1a, over the post-processed image and over the raw sums, rows flipped, as doubles.  The same renders DO
    reproduce Appendix C's iteration tables, event totals and mean RGB to all nine printed digits (test_image_goldens_matte),
    so what differs is the hash definition or ulp-level libm output of the survey's session, neither recoverable here.
    This test keeps the comparison visible: it is an expected failure, and an XPASS would mean the definition was found."""
    from oracle.oracle import fnv1a64_words
    w, h = rec["w"], rec["h"]
    img, _, _ = oracle_libm.scene(bunny_matte).render(default_camera(oracle_libm, w / h), w, h, rec["spp"], threads=1)
    assert "%016x" % fnv1a64_words(img) == rec["fnv_recorded_by_survey"]


@pytest.mark.parametrize("variant", ["full_bsdf", "four_bunnies", "sixteen_lights"])
def test_image_goldens_variants(oracle_libm, variant):
    from rtcuda_amd import scenes
    rec = APPX["images_variants_480x270x4"][variant]
    sc = oracle_libm.scene(scenes.cornell_bunny(variant))
    img, _, st = sc.render(default_camera(oracle_libm, 480 / 270), 480, 270, 4, threads=8, collect_stats=True)
    assert (st["sum_mat"], st["sum_ah"], st["sum_ch"]) == (rec["mat"], rec["ah"], rec["ch"])
    assert np.allclose(img.reshape(-1, 3).mean(0, dtype=np.float64), rec["mean"], rtol=0, atol=6e-10)
    assert st["ch_adds"] == 0
    tr = APPX["traversal_per_ray_480x270x4"][variant]
    assert abs(st["ch_node_pairs"] / st["ch_rays"] - tr["np_c"]) < 0.006
    assert abs(st["ch_tri_tests"] / st["ch_rays"] - tr["tt_c"]) < 0.006
    assert abs(st["ah_node_pairs"] / st["ah_rays"] - tr["np_a"]) < 0.006
    assert abs(st["ah_tri_tests"] / st["ah_rays"] - tr["tt_a"]) < 0.006
    assert st["max_stack"] <= 14


def test_xorwow_step_and_subsequence_jump_against_rocrands_engine(oracle, tmp_path):
    """An implementation of the third-party generator that this project did not write: rocRAND's `xorwow_engine` (ROCm image,
    /opt/rocm/include/rocrand/rocrand_xorwow.h) is the same XORWOW -- same xorshift step, same Weyl increment, same 2^67-draw
    subsequences applied with ITS OWN precomputed GF(2) matrices -- and differs from cuRAND only in the seed-scramble
    constants.  Started on the host from the oracle's cuRAND-scrambled state of a seed (tests/cpp/xorwow_rocrand_check.cpp), it
    must reach the oracle's curand_init(seed, k, 0) state and draws for every k tried.  What stays unverifiable offline is
    the scramble itself: five constants."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc or not os.path.exists("/opt/rocm/include/rocrand/rocrand_xorwow.h"):
        pytest.skip("no hipcc / rocRAND headers in this environment")
    exe = str(tmp_path / "xorwow_rocrand_check")
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I/opt/rocm/include",
                           os.path.join(HERE, "cpp", "xorwow_rocrand_check.cpp"), "-o", exe], stderr=subprocess.DEVNULL)
    ks = [0, 1, 2, 3, 5, 64, 4095, 12345, 524288, 1048575]
    for seed in (1, 12345, 0xDEADBEEFCAFE):
        base = oracle.xorwow_init(seed, 0)
        out = subprocess.run([exe] + [str(int(x)) for x in base] + ["4"] + [str(k) for k in ks], capture_output=True,
                             text=True, check=True).stdout.splitlines()
        assert len(out) == len(ks)
        for line, k in zip(out, ks):
            t = [int(x) for x in line.split()]
            assert t[0] == k
            mine = oracle.xorwow_init(seed, k)
            assert np.array_equal(np.array(t[1:7], np.uint32), mine), (seed, k)
            raw, _ = oracle.xorwow_draw(mine.copy(), 4)
            assert [int(x) for x in raw] == t[7:], (seed, k)


def test_pinned_math_accuracy(oracle):
    """rt_sincosf / rt_pow5f (the pinned stand-ins for sincosf / powf(x,5)) are within 2 ulp on their ranges."""
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 20001), [1e-7, np.pi / 2, np.pi, 6.2831855]]).astype(np.float32)
    worst = 0.0
    for x in xs[::7]:
        s, c = oracle.sincos(float(x))
        for got, ref in ((s, np.sin(np.float64(x))), (c, np.cos(np.float64(x)))):
            ulp = max(np.spacing(np.float32(abs(ref))), np.float32(2 ** -24))
            worst = max(worst, abs(got - ref) / ulp)
    assert worst <= 2.0, worst
    for x in np.linspace(0, 1, 101).astype(np.float32):
        ref = np.float64(x) ** 5
        assert abs(oracle.pow5(float(x)) - ref) <= 2.5 * np.spacing(np.float32(ref)) + 1e-45


def test_pinned_and_libm_flavours_agree_statistically(oracle, oracle_libm, bunny_matte):
    """Swapping sincosf for the pinned polynomial re-draws a few diffuse directions by an ulp: event
    totals move by a handful, the image by far less than the Monte-Carlo noise."""
    w, h, spp = 64, 36, 8
    a, _, sa = oracle.scene(bunny_matte).render(default_camera(oracle, w / h), w, h, spp, threads=8)
    b, _, sb = oracle_libm.scene(bunny_matte).render(default_camera(oracle_libm, w / h), w, h, spp, threads=8)
    assert abs(sa["sum_mat"] - sb["sum_mat"]) <= 0.002 * sb["sum_mat"]
    assert abs(a.mean(dtype=np.float64) - b.mean(dtype=np.float64)) < 2e-3


def test_oracle_shards_sum_to_full_image(oracle, bunny_matte):
    """Partition invariance on the CPU: slot s serves camera rays c == s (mod W), so slot-range shards
    are disjoint and their raw sums add up to the unsharded image."""
    w, h, spp = 48, 27, 8
    sc = oracle.scene(bunny_matte)
    cam = default_camera(oracle, w / h)
    _, full, st_full = sc.render(cam, w, h, spp, threads=8)
    W = 1 << 20
    acc = np.zeros_like(full)
    mats = 0
    for r in range(2):
        _, part, st = sc.render(cam, w, h, spp, slot_lo=r * W // 2, slot_hi=(r + 1) * W // 2, threads=8)
        acc += part
        mats += st["sum_mat"]
    assert mats == st_full["sum_mat"]
    assert np.allclose(acc, full, rtol=1e-5, atol=1e-6)


# ---- committed render fixtures (tests/golden/make_golden.py): the oracle still produces them, in BOTH of its modes
GOLDEN = np.load(os.path.join(HERE, "golden", "render_goldens.npz"))
AUDITED = ("sum_mat", "sum_ah", "emission_adds", "ah_adds", "rr_draws")


def test_literal_and_watertight_fixtures_are_reproduced(oracle):
    """`literal_*`: the reference's own slab test and tie rule (default mode) -- bit for bit.  The same frame in watertight
    mode (what the strict GPU comparisons use) may differ from it only by the audited handful of rays (the reference's
    walk loses about one accepted hit in 10^7: tests/test_traversal_audit.py), so neither mode can drift unnoticed."""
    from rtcuda_amd import scenes
    variant, w, h, spp = "matte", 160, 100, 160
    key = f"literal_{variant}_{w}x{h}x{spp}"
    sc = oracle.scene(scenes.cornell_bunny(variant))
    cam = default_camera(oracle, w / h)
    img, _, st = sc.render(cam, w, h, spp, threads=usable_cpus())
    assert [st[k] for k in AUDITED] + [w * h * spp] == GOLDEN[key + "_counts"].tolist()
    assert np.array_equal(img.view(np.uint32), GOLDEN[key + "_img"].view(np.uint32))
    img_w, _, st_w = sc.set_watertight(True).render(cam, w, h, spp, threads=usable_cpus())
    for k, want in zip(AUDITED, GOLDEN[key + "_counts"].tolist()):
        assert abs(st_w[k] - want) <= 4, (k, st_w[k], want)
    d = np.abs(img_w.astype(np.float64) - GOLDEN[key + "_img"])
    assert (d.max(axis=2) > 1e-4).sum() <= 2 and np.sqrt(np.mean(d ** 2)) < 1e-4


def test_full_size_totals_fixture_is_well_formed():
    """tests/golden/full_size_event_totals.json (the oracle's totals of the six full BASELINE frames, which the GPU suite
    holds k_paths against): one entry per BASELINE frame, sample counts consistent, literal and watertight totals within
    the audited distance of each other (about one ray in 10^7)."""
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    frames = json.load(open(os.path.join(here, "golden", "full_size_event_totals.json")))["frames"]
    assert sorted((f["scene"], f["spp"]) for f in frames) == [("four_bunnies", 256), ("full_bsdf", 256), ("matte", 256), ("matte", 1024),
                                                              ("sixteen_lights", 256), ("sixteen_lights", 512)]
    for f in frames:
        assert f["samples"] == f["width"] * f["height"] * f["spp"] and (f["width"], f["height"]) == (1920, 1080)
        for k, v in f["oracle_watertight"].items():
            assert v > 0 and abs(v - f["oracle_literal"][k]) <= 2e-6 * f["oracle_watertight"]["shade_events"], (f["scene"], k)
