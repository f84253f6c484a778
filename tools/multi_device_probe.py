#!/usr/bin/env python3
"""One frame through rt_render_multi (one process, one host thread per GPU) in a process of its own; prints one JSON line.

bench.py's rank 0 runs this as a CHILD process while the other ranks idle, so that the in-process multi-device path gets
exercised on the node's real GPUs without putting the bench line at risk: whatever happens in here -- an error, a hang (the
caller's timeout), a crash -- costs a sub-record, not the measurement.

usage: multi_device_probe.py scene width height spp max_bounces dev0,dev1,..."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    scene_name, w, h, spp, max_bounces = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    devices = [int(x) for x in sys.argv[6].split(",")]
    import numpy as np
    from rtcuda_amd import api, scenes
    sc = api.Scene(scenes.cornell_bunny(scene_name))  # on device 0 of this process; replicas are made by the first call
    cam = api.make_camera(aspect=w / h)
    t0 = time.perf_counter()
    sc.render_multi(cam, w, h, spp, devices, max_bounces=max_bounces, seed=1)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    img, st = sc.render_multi(cam, w, h, spp, devices, max_bounces=max_bounces, seed=1)
    t_second = time.perf_counter() - t0
    keys = ("camera_rays", "shade_events", "any_rays", "emission_adds", "shadow_adds", "rr_draws", "closest_rays", "literal_retraces",
            "reference_lost_hits", "exact_ties")
    print(json.dumps({"devices": devices, "device_shards": st["device_shards"], "wall_ms_first_call": round(1e3 * t_first, 2),
                      "wall_ms": round(1e3 * t_second, 2), "device_ms_slowest_shard": round(1e3 * st["seconds_render"], 3),
                      "Msamples_per_s_wall": round(float(w) * h * spp / t_second / 1e6, 1),
                      "totals": {k: int(st[k]) for k in keys}, "peer_access": api.peer_access_log(),
                      "image_mean": float(np.nanmean(img, dtype=np.float64)),
                      "nan_pixels": int(np.isnan(img).any(axis=2).sum())}), flush=True)


if __name__ == "__main__":
    main()
