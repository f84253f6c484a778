#!/usr/bin/env python3
"""Regenerates tests/golden/ref_host_fixture.npz from the REFERENCE'S OWN host code (only where /root/reference exists).

`make -C oracle _ref_host` compiles oracle/ref_host_driver.cpp against /root/reference/happly.h, matrix4x4.hpp and
transform.hpp -- included by path, unmodified: the three files of the reference that are pure host C++ -- into
oracle/_ref/ref_host (git-ignored).  The driver performs the scene-preparation calls of main.cu:59-71 and writes numbers;
this script stores them as the fixture.  Nothing of the reference's text is stored: vertex positions, index triples,
matrices.

What the fixture pins (tests/test_host_api_cpp.py, tests/test_host_logic.py): include/rtcuda/ply.hpp, matrix4x4.hpp,
transform.hpp (the same driver source compiled against them must write the same bytes) and rtcuda_amd/scenes.py
(load_ply, bunny_transform, composite, apply_transform, rotate) -- i.e. SURVEY section 8 rows f1 and f2, and with them
the geometry every parity test and the benchmark render.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REF", "/root/reference")


def read_dump(path):
    """The binary layout oracle/ref_host_driver.cpp writes -> dict of arrays."""
    b = open(path, "rb").read()
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(b, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a.copy()
    nv, nf = (int(x) for x in take(np.int64, 2))
    out = {"parsed_positions_f64": take(np.float64, 3 * nv).reshape(nv, 3), "faces": take(np.int32, 3 * nf).reshape(nf, 3),
           "bunny_matrix": take(np.float32, 16).reshape(4, 4), "transformed_f32": take(np.float32, 3 * nv).reshape(nv, 3)}
    nk = int(take(np.int64, 1)[0])
    out["applied_every_997th_f64"] = take(np.float64, 3 * nk).reshape(nk, 3)
    nr = int(take(np.int64, 1)[0])
    rec = take(np.float32, 20 * nr).reshape(nr, 20)
    out["rotate_args"], out["rotate_matrices"] = rec[:, :4].copy(), rec[:, 4:].reshape(nr, 4, 4).copy()
    na = int(take(np.int64, 1)[0])
    out["rotated_applies_f64"] = take(np.float64, 3 * na).reshape(na, 3)
    assert off == len(b), (off, len(b))
    return out


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} does not exist: the fixture can only be regenerated where the reference is present")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref_host", f"REF={REF}"])
    tmp = os.path.join(ROOT, "oracle", "_ref", "ref_host.bin")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_host"), os.path.join(REF, "bun_zipper.ply"), tmp])
    d = read_dump(tmp)
    p32 = d.pop("parsed_positions_f64").astype(np.float32)
    assert np.array_equal(p32.astype(np.float64), read_dump(tmp)["parsed_positions_f64"])  # happly: text -> fp32 -> double
    d["parsed_positions_f32"] = p32
    out = os.path.join(HERE, "ref_host_fixture.npz")
    np.savez_compressed(out, **d)
    print(out, os.path.getsize(out), "bytes;", {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
