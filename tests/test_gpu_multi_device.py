"""rt_render_multi -- several GPUs of one node behind ONE call in ONE process (C-ABI, C++ render() overload).  -m gpu.

No reference counterpart: render() (render.cuh:366-367, called once from main.cu:173) drives one device from one thread.
The path is: scene replicated per device, one host thread per device, slot-range shards (the partition the N-process path
uses: rtcuda_amd/dist.py), the shards' raw sums copied to devices[0] and added in shard order, post-process there.
A device may be listed more than once -- its shards then run side by side on it -- which is what a one-GPU box can test:
every line of the path except the peer copy between two different devices runs here.

Bar: in RT_FLAG_DETERMINISTIC arithmetic the image is BIT-EQUAL to rt_render's whatever the device list; event totals equal;
in the default (float atomics) arithmetic RMS < 2e-6.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = ("camera_rays", "shade_events", "closest_rays", "any_rays", "emission_adds", "shadow_adds", "rr_draws")


@pytest.fixture(scope="module")
def api():
    from rtcuda_amd import api as _api
    _api.lib()
    return _api


@pytest.fixture(scope="module")
def gpu_full(api):
    from rtcuda_amd import scenes
    sc = api.Scene(scenes.cornell_bunny("full_bsdf"))
    yield sc
    sc.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0], [0] * 8])
def test_multi_device_render_is_bit_equal_to_the_single_device_render(api, gpu_full, devices):
    """3.96 generations of the full-BSDF scene.  [0] * 8 launches the small-shard build of k_paths eight times side by side."""
    w, h, spp = 480, 270, 32
    cam = api.make_camera(aspect=w / h)
    ref, st_ref = gpu_full.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    img, st = gpu_full.render_multi(cam, w, h, spp, devices, flags=api.FLAG_DETERMINISTIC)
    assert st["device_shards"] == len(devices)
    for k in KEYS:
        assert st[k] == st_ref[k], k
    assert st["camera_rays"] == w * h * spp
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    # default arithmetic (float atomics per shard, float adds in shard order on devices[0])
    img_f, st_f = gpu_full.render_multi(cam, w, h, spp, devices)
    ref_f, _ = gpu_full.render(cam, w, h, spp)
    assert all(st_f[k] == st_ref[k] for k in KEYS)
    nan = np.isnan(ref_f)
    assert np.array_equal(np.isnan(img_f), nan)
    d = np.where(nan, 0.0, img_f.astype(np.float64) - np.where(nan, 0.0, ref_f))
    assert np.sqrt(np.mean(d ** 2)) < 2e-6


def test_multi_device_render_in_the_other_modes(api, gpu_full):
    """RT_FLAG_REFERENCE_WALK (each replica... here: the one scene builds the reference's tree once) and RT_FLAG_RNG_PER_SAMPLE
    (every shard renders ALL slots at spp / n) through the same entry point, each bit-equal to its single-device render."""
    w, h, spp = 300, 200, 48
    cam = api.make_camera(aspect=w / h)
    for extra in (api.FLAG_REFERENCE_WALK, api.FLAG_RNG_PER_SAMPLE, api.FLAG_WATERTIGHT):
        F = api.FLAG_DETERMINISTIC | extra
        ref, st_ref = gpu_full.render(cam, w, h, spp, flags=F)
        img, st = gpu_full.render_multi(cam, w, h, spp, [0, 0, 0, 0], flags=F)
        assert all(st[k] == st_ref[k] for k in KEYS), extra
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), extra


def test_multi_device_render_rejects_bad_device_lists(api, gpu_full):
    cam = api.make_camera(aspect=1.0)
    for bad in ([0, 0, 0], [99], [-1], []):
        with pytest.raises(api.RtError):
            gpu_full.render_multi(cam, 32, 32, 4, bad)
    with pytest.raises(api.RtError):  # per-sample streams split the SAMPLES over the shards
        gpu_full.render_multi(cam, 32, 32, 5, [0, 0], flags=api.FLAG_RNG_PER_SAMPLE)
    img, st = gpu_full.render_multi(cam, 32, 32, 4, [0, 0])  # ... and the scene is still usable
    assert st["camera_rays"] == 32 * 32 * 4 and np.isfinite(img).all()


def test_cpp_driver_reaches_several_devices_through_render(api, tmp_path):
    """examples/cornell_bunny --devices 0,0 (the render() overload with a device list) and RTCUDA_DEVICES=0,0 (the
    reference's unchanged seven-argument call, main.cu:173) write the PPM of the single-device run."""
    exe = os.path.join(ROOT, "examples", "cornell_bunny")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "rtcuda_amd", "csrc"), "example"])
    w, h, spp = 96, 96, 12  # (one generation: 110 592 camera rays)
    ply = os.path.join(ROOT, "data", "bun_zipper.ply")

    def run(name, pre=(), env=None):
        out = tmp_path / name
        log = subprocess.run([exe, *pre, str(w), str(h), str(spp), ply, str(out), "full_bsdf"], cwd=ROOT, check=True,
                             capture_output=True, text=True, env=env).stdout
        tok = out.read_text().split()
        assert tok[:4] == ["P3", str(w), str(h), "255"]
        return np.array(tok[4:], np.int64).reshape(h, w, 3), log
    one, _ = run("one.ppm")
    two, log2 = run("two.ppm", pre=("--devices", "0,0"))
    assert "rendered as 2 device shard(s)" in log2
    env = dict(os.environ, RTCUDA_DEVICES="0,0,0,0")
    four, _ = run("four.ppm", env=env)
    for got in (two, four):  # identical contributions; the order of the float adds may flip a quantisation boundary
        assert (got != one).mean() < 1e-3 and np.abs(got - one).max() <= 1
    # the library's modes behind the unchanged call: RTCUDA_DETERMINISTIC=1 is bit-reproducible (the same PPM twice, and from
    # two device shards), RTCUDA_REFERENCE_WALK=1 renders the same scene through the reference's own tree
    det_env = dict(os.environ, RTCUDA_DETERMINISTIC="1")
    d1, _ = run("det1.ppm", env=det_env)
    d2, _ = run("det2.ppm", env=dict(det_env, RTCUDA_DEVICES="0,0"))
    assert np.array_equal(d1, d2)
    assert (d1 != one).mean() < 1e-3 and np.abs(d1 - one).max() <= 1
    ref, _ = run("ref.ppm", env=dict(os.environ, RTCUDA_REFERENCE_WALK="1"))
    assert (ref != one).mean() < 1e-2 and ref.any()


def test_shutdown_releases_the_hidden_allocations_and_the_library_keeps_working(api, gpu_full):
    """rt_shutdown frees the render contexts (path pools, RNG states, counters, overflow stacks) and the cached output buffers
    of rt_render / rt_render_multi; the next call re-creates what it needs and renders the same image."""
    import torch
    w, h, spp = 96, 54, 8
    cam = api.make_camera(aspect=w / h)
    a, st_a = gpu_full.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    m, _ = gpu_full.render_multi(cam, w, h, spp, [0, 0], flags=api.FLAG_DETERMINISTIC)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    api.shutdown()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_after >= free_before + (100 << 20)  # the full-pool context alone is 36 x 4 MB of path state + 24 MB of RNG backup
    b, st_b = gpu_full.render(cam, w, h, spp, flags=api.FLAG_DETERMINISTIC)
    m2, _ = gpu_full.render_multi(cam, w, h, spp, [0, 0], flags=api.FLAG_DETERMINISTIC)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(m.view(np.uint32), m2.view(np.uint32))
    assert st_a["shade_events"] == st_b["shade_events"]
    api.shutdown()
    api.shutdown()  # (idempotent)


def test_peer_access_log_is_empty_until_two_different_devices_are_listed(api, gpu_full):
    """rt_render_multi enables peer access devices[0] <-> devices[k] once per pair and says what became of it
    (rt_peer_access_log).  A one-GPU box can only list device 0 several times: nothing to enable, an empty log -- the copy
    between two physical devices runs for the first time in the driver's multi-GPU run, where bench.py records this string."""
    cam = api.make_camera(aspect=1.0)
    gpu_full.render_multi(cam, 32, 32, 4, [0, 0])
    import torch
    if torch.cuda.device_count() == 1:
        assert api.peer_access_log() == ""
    else:
        gpu_full.render_multi(cam, 32, 32, 4, [0, 1])
        assert "devices 0 <-> 1" in api.peer_access_log()
