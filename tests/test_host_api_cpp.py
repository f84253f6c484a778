"""CPU tests of the C++ host boundary (include/rtcuda/*.hpp) -- SURVEY.md section 8 b, f1, f2, f3.

tests/cpp/host_api_check.cpp is compiled against the shippable headers and the product library and run in its
host-only modes (no GPU call is made): the scene recipes, the PLY reader, the transforms and the Vec3 arithmetic of
the C++ side must give the SAME BYTES as rtcuda_amd/scenes.py, which feeds the oracle and the GPU tests.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from rtcuda_amd import scenes

EXE = os.path.join(ROOT, "tests", "cpp", "host_api_check")
PLY = os.path.join(ROOT, "data", "bun_zipper.ply")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "rtcuda_amd", "csrc"), "hostcheck"],
                          stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)
    return EXE


def _read_dump(path):
    raw = open(path, "rb").read()
    n, nl = struct.unpack_from("<qq", raw, 0)
    off = 16
    tris = np.frombuffer(raw, np.float32, n * 9, off).reshape(n, 9)
    off += n * 36
    mat = np.frombuffer(raw, np.int32, n, off)
    off += 4 * n
    lig = np.frombuffer(raw, np.int32, n, off)
    off += 4 * n
    lights = np.frombuffer(raw, np.dtype([("tri", "<i4"), ("L", "<f4", 3)]), nl, off)
    off += 16 * nl
    mats = np.frombuffer(raw, np.dtype([("albedo", "<f4", 3), ("ior", "<f4"), ("type", "<i4")]), 6, off)
    assert off + 6 * 20 == len(raw)
    return tris, mat, lig, lights, mats


@pytest.mark.parametrize("variant", ["matte", "full_bsdf", "four_bunnies", "sixteen_lights"])
def test_cpp_scene_recipe_equals_python_recipe_bit_for_bit(exe, tmp_path, variant):
    out = str(tmp_path / "scene.bin")
    subprocess.check_call([exe, "dump", variant, PLY, out])
    tris, mat, lig, lights, mats = _read_dump(out)
    ref = scenes.cornell_bunny(variant)
    assert tris.shape == ref.tris.shape
    assert np.array_equal(tris.view(np.uint32), ref.tris.view(np.uint32))  # every vertex, every bit
    assert np.array_equal(mat, ref.tri_material)
    assert np.array_equal(lig, ref.tri_light)
    assert np.array_equal(lights["tri"], ref.lights["tri"])  # light ORDER
    assert np.array_equal(lights["L"], ref.lights["L"])
    assert np.array_equal(mats["albedo"], ref.materials["albedo"])
    assert np.array_equal(mats["ior"], ref.materials["ior"]) and np.array_equal(mats["type"], ref.materials["type"])


def test_cpp_bare_box(exe, tmp_path):
    out = str(tmp_path / "box.bin")
    subprocess.check_call([exe, "dump", "matte", "-", out])
    tris, mat, lig, lights, _ = _read_dump(out)
    ref = scenes.cornell_bunny("matte", bunny=False)
    assert np.array_equal(tris.view(np.uint32), ref.tris.view(np.uint32))
    assert np.array_equal(mat, ref.tri_material) and np.array_equal(lig, ref.tri_light)
    assert lights["tri"].tolist() == [11, 10]


def _read_ply_dump(path):
    raw = open(path, "rb").read()
    nv, nf = struct.unpack_from("<qq", raw, 0)
    pos = np.frombuffer(raw, np.float64, nv * 3, 16).reshape(nv, 3)
    off = 16 + nv * 24
    faces = []
    for _ in range(nf):
        (k,) = struct.unpack_from("<q", raw, off)
        faces.append(np.frombuffer(raw, np.int64, k, off + 8).tolist())
        off += 8 + 8 * k
    assert off == len(raw)
    return pos, faces


def _write_binary_ply(path, pos, faces, big_endian, double_xyz=False):
    e = ">" if big_endian else "<"
    with open(path, "wb") as fh:
        t = "double" if double_xyz else "float"
        fh.write((f"ply\nformat binary_{'big' if big_endian else 'little'}_endian 1.0\ncomment test\n"
                  f"element vertex {len(pos)}\nproperty {t} x\nproperty {t} y\nproperty {t} z\nproperty uchar flag\n"
                  f"element face {len(faces)}\nproperty list uchar int vertex_indices\nproperty short tag\n"
                  "end_header\n").encode())
        for p in pos:
            fh.write(struct.pack(e + ("3d" if double_xyz else "3f") + "B", *p, 7))
        for f in faces:
            fh.write(struct.pack(e + "B" + f"{len(f)}i" + "h", len(f), *f, -3))


def test_cpp_ply_reader_ascii_and_binary(exe, tmp_path):
    out = str(tmp_path / "ply.bin")
    subprocess.check_call([exe, "ply", PLY, out])
    pos, faces = _read_ply_dump(out)
    ref_pos, ref_faces = scenes.load_ply(PLY)
    assert pos.shape == (35947, 3) and len(faces) == 69451  # bun_zipper.ply:4,10
    # `property float`: text -> fp32 -> double (happly.h:318-325), so every double is exactly an fp32 value
    assert np.array_equal(pos, ref_pos.astype(np.float64))
    assert np.array_equal(np.array(faces), ref_faces)
    # binary, both byte orders, extra properties either side of what is read, a quad among the faces
    rng = np.random.default_rng(3)
    p = rng.normal(size=(50, 3)).astype(np.float32)
    f = [[0, 1, 2], [3, 4, 5, 6], [7, 8, 9]]
    for big in (False, True):
        path = str(tmp_path / f"b{int(big)}.ply")
        _write_binary_ply(path, p.tolist(), f, big)
        subprocess.check_call([exe, "ply", path, out])
        pos, faces = _read_ply_dump(out)
        assert np.array_equal(pos, p.astype(np.float64)) and faces == f
    path = str(tmp_path / "d.ply")
    pd = rng.normal(size=(5, 3))
    _write_binary_ply(path, pd.tolist(), f[:1], False, double_xyz=True)
    subprocess.check_call([exe, "ply", path, out])
    pos, _ = _read_ply_dump(out)
    assert np.array_equal(pos, pd)  # `property double` is not squeezed through fp32


def test_cpp_ply_reader_rejects_malformed_files(exe, tmp_path):
    out = str(tmp_path / "o.bin")
    bad = tmp_path / "bad.ply"
    bad.write_text("plx\nformat ascii 1.0\nend_header\n")
    assert subprocess.call([exe, "ply", str(bad), out], stderr=subprocess.DEVNULL) == 1
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
                   "element face 0\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n")  # truncated body
    assert subprocess.call([exe, "ply", str(bad), out], stderr=subprocess.DEVNULL) == 1
    assert subprocess.call([exe, "ply", str(tmp_path / "missing.ply"), out], stderr=subprocess.DEVNULL) == 1


def test_cpp_vec3_matrix_transform_known_answers(exe):
    lines = subprocess.check_output([exe, "unit"], text=True).strip().splitlines()
    got = {k: [float(x) for x in v.split()] for k, v in (ln.split("=") for ln in lines)}
    f = np.float32
    a, b = np.array([1, -2, 3], f), np.array([0.5, 4, -0.25], f)

    def eq(name, ref):
        assert np.array_equal(np.array(got[name], f), np.atleast_1d(np.asarray(ref, f))), (name, got[name], ref)

    eq("add", a + b); eq("sub", a - b); eq("mul", a * b); eq("div", a / b)
    eq("scale_l", f(3) * a); eq("scale_r", a * f(3)); eq("neg", -a)
    eq("div_s", a * (f(1) / f(3)))  # reciprocal-multiply, NOT a / 3 (vec3.cuh:56-59)
    a7 = np.array([5, 9, 13], f)
    eq("div_s7", a7 * (f(1) / f(7)))
    assert not np.array_equal(a7 * (f(1) / f(7)), a7 / f(7))  # (the two forms really differ for this input)
    dot = f(f(a[0] * b[0]) + f(a[1] * b[1])) + f(a[2] * b[2])
    eq("dot", dot)
    eq("cross", [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])
    len2 = f(f(a[0] * a[0]) + f(a[1] * a[1])) + f(a[2] * a[2])
    eq("len2", len2); eq("len", np.sqrt(len2)); eq("unit", a * (f(1) / np.sqrt(len2))); eq("max", 3)
    eq("reflect", [0.6, 0.8, 0.0])
    eq("sqrt", np.sqrt(np.array([4, 9, 2], f))); eq("zeros_ones", [1, 1, 1])
    c = a + b
    c = c - f(1); c = c * np.array([2, 3, 4], f); c = c / f(2); c = c * f(0.5); c = c * (f(1) / f(3))
    eq("compound", c)
    # refract (vec3.cuh:76-86): eta narrows to fp32 at the scalar * Vec3 product
    v, n, eta = np.array([0.6, -0.8, 0], f), np.array([0, 1, 0], f), f(1.0 / 1.5)
    cos_t = -(f(f(v[0] * n[0]) + f(v[1] * n[1])) + f(v[2] * n[2]))
    par = eta * (v + cos_t * n)
    par2 = f(f(par[0] * par[0]) + f(par[1] * par[1])) + f(par[2] * par[2])
    eq("refract", par + (-np.sqrt(f(1) - par2)) * n)
    # Matrix4x4::Rotate, Transform::composite / ::apply against the numpy restatement in scenes.py
    s3 = f(0.577350259)
    rot = scenes.rotate(s3, s3, s3, 0.7)
    for i in range(3):
        assert np.allclose(np.array(got[f"rot{i}"], f), rot[i], rtol=0, atol=6e-8), (i, got[f"rot{i}"], rot[i])  # cosf/sinf: libm vs numpy, <= 1 ulp
    m = scenes.bunny_transform()
    for i in range(3):
        eq(f"bunny{i}", m[i])  # [2,0,0,0.489379823] [.., -0.0659748018] [.., -0.617599368]
    assert got["bunny0"][3] == pytest.approx(0.489379823, abs=1e-9)
    rot_c = np.array([got[f"rot{i}"] for i in range(3)] + [[0, 0, 0, 1]], f)  # the C++ side's own rotation matrix
    m2 = scenes.composite(m, rot_c)
    p = np.array([[-0.0378297, 0.12794, 0.00447467]])
    md = m2.astype(np.float64)
    ref = [(((md[i, 0] * p[0, 0] + md[i, 1] * p[0, 1]) + md[i, 2] * p[0, 2]) + md[i, 3]) for i in range(3)]
    assert got["applied"][0] == float(f(ref[0])) and got["applied"][1] == float(f(ref[1]))  # x, y rounded to fp32
    assert got["applied"][2] == ref[2]                                                       # z kept in double
    assert got["point_light"] == [0.0, 1.0]
    # ---- host PODs (row a30): Triangle members / accessors, BoundingBox, Intersection, Ray
    p0, p1, p2 = np.array([0.4, 0.999, -0.4], f), np.array([0.6, 0.999, -0.4], f), np.array([0.6, 0.999, -0.6], f)
    e1, e2 = p0 - p1, p2 - p0
    n = np.array([e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]], f)
    eq("tri_e1", e1); eq("tri_e2", e2); eq("tri_n", n)
    assert f(got["tri_e1"][0]) == f(-0.200000018) and f(got["tri_n"][1]) == f(-0.0400000066)
    eq("tri_p1", p0 - e1); eq("tri_p2", p0 + e2)  # the reference's rounded reconstructions (triangle.cuh:9-10)
    eq("tri_center", ((p0 + (p0 - e1)) + (p0 + e2)) * (f(1) / f(3)))
    eq("tri_puv", (p0 - f(0.25) * e1) + f(0.5) * e2)
    nl2 = f(f(n[0] * n[0]) + f(n[1] * n[1])) + f(n[2] * n[2])
    eq("tri_area", f(0.5) * np.sqrt(nl2))
    assert f(got["tri_area"][0]) == f(0.0200000033)
    q1, q2 = p0 - e1, p0 + e2
    eq("tri_bbox", [min(p0[0], q1[0], q2[0]), max(p0[0], q1[0], q2[0]), min(p0[1], q1[1], q2[1]),
                    max(p0[1], q1[1], q2[1]), min(p0[2], q1[2], q2[2]), max(p0[2], q1[2], q2[2])])
    assert got["bbox_empty"] == [1.0] and got["bbox_reset"] == [1.0] and got["ray_tmax_default"] == [1.0]
    ext = [0.0, 1.0, 0.0, float(max(p0[1], q1[1], q2[1])), -1.0, 0.0]
    eq("bbox_ext", ext)
    dx, dy, dz = f(ext[1]) - f(ext[0]), f(ext[3]) - f(ext[2]), f(ext[5]) - f(ext[4])
    eq("bbox_half_area", f(f(dx + dy) * dz) + f(dx * dy))
    eq("ray_at", np.array([0.55, 0.2, -0.45], f) + f(0.799000025) * np.array([0, 1, 0], f))
    assert np.array(got["offset_a"], f).view(np.uint32).tolist() == [0x3E99999A, 0x3CA3F70A, 0xBF333333]  # Appendix C
    assert np.array(got["offset_b"], f).view(np.uint32).tolist() == [0x3E999A33, 0x3EFFFF34, 0xBF333333]
    assert got["spawn_tmax"] == [2.5]


def test_stage_lines_have_the_reference_drivers_format(exe):
    """profiler.hpp:14-28 prints "<name>... done (<ms>ms)"; the recipe prints the driver's three bunny stages
    (main.cu:59-85) in that format when asked to, and nothing when not."""
    import re
    out = subprocess.check_output([exe, "stages", PLY], text=True).splitlines()
    stage = re.compile(r"^(.+)\.\.\. done \(([0-9.eE+-]+)ms\)$")
    names = [stage.match(ln).group(1) for ln in out if stage.match(ln)]
    assert names == ["Reading bunny", "Transforming bunny", "Converting bunny to triangles", "Explicit stage", "Scoped stage"]
    assert out[1] == "35947 vertices, 69451 faces"  # (the count line the driver prints after reading: main.cu:63)
    assert out[-1] == "triangles=69463 last_ms_ok=1" and not any("Silent" in ln for ln in out)


def test_reference_driver_lines_compile_against_the_headers(tmp_path):
    """A driver written the way main.cu is -- happly::PLYData, Transform, Vec3 arithmetic, Material / Light /
    Primitive / Bvh / Scene / Camera / render() -- compiles (syntax-only: no GPU, no link) against include/."""
    src = tmp_path / "driver.cpp"
    src.write_text(r'''
#define RTCUDA_PLY_AS_HAPPLY
#include <array>
#include <vector>
#include "rtcuda/rtcuda.hpp"
int main() {
    std::vector<Material> materials;
    materials.push_back(Material::make_matte(Vec3(0.65f, 0.05f, 0.05f)));
    materials.push_back(Material::make_mirror(Vec3(0.9f)));
    materials.push_back(Material::make_glass(1.5f));
    happly::PLYData ply_in("../bun_zipper.ply");
    std::vector<std::array<double, 3>> v_pos = ply_in.getVertexPositions();
    std::vector<std::vector<size_t>> f_index = ply_in.getFaceIndices<size_t>();
    Transform transform(Matrix4x4::Translate(0.1f, 0.f, 0.f));
    transform.composite(Matrix4x4::Rotate(0.f, 1.f, 0.f, 0.5f));
    transform.composite(Matrix4x4::Scale(2.f, 2.f, 2.f));
    for (auto &v : v_pos) transform.apply(v);
    std::vector<Triangle> triangles;
    std::vector<Material*> material_ptrs;
    for (int i = 0; i < (int)f_index.size(); i++) {
        const std::vector<size_t> &face = f_index[i];
        triangles.emplace_back(Vec3(v_pos[face[0]][0], v_pos[face[0]][1], v_pos[face[0]][2]),
                               Vec3(v_pos[face[1]][0], v_pos[face[1]][1], v_pos[face[1]][2]),
                               Vec3(v_pos[face[2]][0], v_pos[face[2]][1], v_pos[face[2]][2]));
        material_ptrs.push_back(&materials[i % 3]);
    }
    Vec3 centre = Vec3::make_zeros();
    for (const Triangle &t : triangles) centre += (t.p0 + t.p1_ + t.p2_) / 3.f;
    centre /= (float)triangles.size();
    Vec3 up = cross(Vec3(1.f, 0.f, 0.f), Vec3(0.f, 0.f, -1.f)).unit_vector();
    std::vector<Light> lights;
    lights.push_back(Light::make_point_light(centre + 2.f * up, Vec3(5.f, 5.f, 5.f)));
    lights.push_back(Light::make_area_light(&triangles[0], Vec3(15.f, 15.f, 15.f)));
    std::vector<Primitive> primitives;
    for (int i = 0; i < (int)triangles.size(); i++) {
        if (i == 0) primitives.emplace_back(&triangles[i], material_ptrs[i], &lights[1]);
        else primitives.emplace_back(&triangles[i], material_ptrs[i]);
    }
    Bvh bvh(triangles, primitives);
    Scene scene = { bvh, (int)lights.size(), lights.data() };
    Camera camera(centre - Vec3(0.f, 0.f, -3.f), centre, up, 37.8f, 1.f);
    std::vector<Vec3> framebuffer;
    render(600, 600, 10, 10, camera, scene, framebuffer);
    return dot(framebuffer[0], Vec3::make_ones()) > 0.f;
}
''')
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"), str(src)])


def test_c_abi_header_is_plain_c(tmp_path):
    """include/rtcuda_amd.h is the drop-in boundary: opaque handles, plain pointers and sizes -- it must compile as C
    (a cgo / JNI / N-API binding sees it as C), and a C caller written against it must type-check."""
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include "rtcuda_amd.h"
#include <stddef.h>
int run(const float *verts, int n, const int32_t *tri_mat, const int32_t *tri_light, const rt_material *mats, int n_mats,
        const rt_light *lights, int n_lights, float *rgb, int w, int h) {
    rt_scene *sc = NULL;
    rt_camera cam;
    rt_stats st;
    float from[3] = {0.5f, 0.5f, 1.5f}, at[3] = {0.5f, 0.5f, 0.f}, up[3] = {0.f, 1.f, 0.f};
    if (rt_scene_create(verts, n, tri_mat, tri_light, mats, n_mats, lights, n_lights, &sc)) return 1;
    if (rt_camera_make(from, at, up, 37.8f, (float)w / (float)h, &cam)) return 2;
    if (rt_render(sc, &cam, w, h, 256, 10, 1u, RT_FLAG_DETERMINISTIC, rgb, &st)) return 3;
    rt_scene_destroy(sc);
    return st.camera_rays == (int64_t)w * h * 256 ? 0 : 4;
}
''')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                           str(src)])


# ----------------------------------------------------------------------------- the reference's OWN host code as the pin
def _ref_host_tools():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_host_fixture", os.path.join(ROOT, "tests", "golden", "make_ref_host_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, np.load(os.path.join(ROOT, "tests", "golden", "ref_host_fixture.npz"))


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def test_product_headers_write_what_the_references_own_host_code_wrote(tmp_path):
    """tests/golden/ref_host_fixture.npz holds the OUTPUT OF THE REFERENCE'S OWN FILES: happly.h, matrix4x4.hpp and
    transform.hpp are pure host C++, compile unmodified (oracle/Makefile, target _ref_host) and were run on bun_zipper.ply
    through oracle/ref_host_driver.cpp (tests/golden/make_ref_host_fixture.py).  The SAME driver source compiled against the
    product's own headers (include/rtcuda/ply.hpp, matrix4x4.hpp, transform.hpp) must write the same numbers, bit for bit:
    35 947 parsed positions, 69 451 index triples, the composite bunny matrix (main.cu:68-70), every vertex after
    Transform::apply and the Vec3 narrowing (main.cu:71,79-81), 12 Matrix4x4::Rotate matrices, 48 composite + apply results.
    This is SURVEY section 8 rows f1 / f2 pinned by the reference itself, not by a restatement."""
    mod, fx = _ref_host_tools()
    exe = str(tmp_path / "ref_host_product")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-DREF_HOST_PRODUCT", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "oracle", "ref_host_driver.cpp"), "-o", exe])
    out = str(tmp_path / "product.bin")
    subprocess.check_call([exe, PLY, out])
    got = mod.read_dump(out)
    assert _same_bits(got["parsed_positions_f64"], fx["parsed_positions_f32"].astype(np.float64))
    for k in ("faces", "bunny_matrix", "transformed_f32", "applied_every_997th_f64", "rotate_args", "rotate_matrices", "rotated_applies_f64"):
        assert _same_bits(got[k], fx[k]), k
    assert got["faces"].shape == (69451, 3) and got["transformed_f32"].shape == (35947, 3)  # bun_zipper.ply:4,10


def test_python_scene_recipe_against_the_references_own_host_code():
    """rtcuda_amd/scenes.py (the recipe every parity test and bench.py renders) against the same fixture: load_ply, the
    composite bunny transform, apply_transform + narrowing -- bit for bit; rotate within an ulp of the reference's cosf /
    sinf (numpy's fp32 cos / sin are not glibc's), composite + apply on the reference's own rotation matrices exactly."""
    _, fx = _ref_host_tools()
    pos, faces = scenes.load_ply(PLY)
    assert _same_bits(pos.astype(np.float32), fx["parsed_positions_f32"])
    assert np.array_equal(np.asarray(faces), fx["faces"])
    m = scenes.bunny_transform()
    assert _same_bits(m.astype(np.float32), fx["bunny_matrix"])
    vt = scenes.apply_transform(m, pos.astype(np.float64))
    assert _same_bits(vt, fx["transformed_f32"])
    # the z a Transform::apply leaves is a double; x and y went through fp32 (transform.hpp:27-32)
    kept = fx["applied_every_997th_f64"]
    assert np.array_equal(kept[:, :2], vt[::997, :2].astype(np.float64)) and _same_bits(kept[:, 2].astype(np.float32), vt[::997, 2])
    for args, want in zip(fx["rotate_args"], fx["rotate_matrices"]):
        assert np.allclose(scenes.rotate(*args), want, rtol=0, atol=6e-8), args
    pts = np.array([[-0.0378297, 0.12794, 0.00447467], [0.0, 0.0, 0.0], [1.0, -2.0, 3.0], [0.061, 0.1871, -0.0588]])
    for k, rot in enumerate(fx["rotate_matrices"]):
        got = scenes.apply_transform(scenes.composite(m, rot), pts)
        want = fx["rotated_applies_f64"][4 * k:4 * k + 4]
        assert _same_bits(got, want.astype(np.float32)), k
        assert np.array_equal(want[:, :2], want[:, :2].astype(np.float32).astype(np.float64))  # x, y: fp32 values


def test_reference_host_build_reproduces_the_committed_fixture(tmp_path):
    """Where the reference is present (this container; never the GPU box): rebuild oracle/_ref/ref_host from the reference's
    files where they lie, run it, and compare with the committed fixture -- the fixture is what the recipe says it is."""
    ref = os.environ.get("REF", "/root/reference")
    if not os.path.exists(os.path.join(ref, "happly.h")):
        pytest.skip("the reference tree is not present here")
    mod, fx = _ref_host_tools()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref_host", f"REF={ref}"], stdout=subprocess.DEVNULL)
    out = str(tmp_path / "ref.bin")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_host"), os.path.join(ref, "bun_zipper.ply"), out])
    got = mod.read_dump(out)
    assert _same_bits(got["parsed_positions_f64"], fx["parsed_positions_f32"].astype(np.float64))
    for k in ("faces", "bunny_matrix", "transformed_f32", "applied_every_997th_f64", "rotate_args", "rotate_matrices", "rotated_applies_f64"):
        assert _same_bits(got[k], fx[k]), k


def test_rtcuda_devices_is_parsed_strictly(exe):
    """RTCUDA_DEVICES sends the reference's unchanged render() call through rt_render_multi.  A typo must not silently change
    which GPUs render: anything but non-negative ordinals separated by single commas throws ('a,b' used to give a
    single-device render, '0,x' the list [0])."""
    def run(value):
        env = dict(os.environ)
        env.pop("RTCUDA_DEVICES", None)
        if value is not None:
            env["RTCUDA_DEVICES"] = value
        return subprocess.run([exe, "devices"], capture_output=True, text=True, env=env)
    assert run(None).stdout.split() == []
    assert run("0").stdout.split() == ["0"]
    assert run("0,1,2,3").stdout.split() == ["0", "1", "2", "3"]
    assert run("3,3").stdout.split() == ["3", "3"]
    for bad in ("a,b", "0,x", "0,,1", "0,1,", ",0", "", "-1", "0 1", "1;2"):
        p = run(bad)
        assert p.returncode == 1 and "RTCUDA_DEVICES" in p.stderr, (bad, p.returncode, p.stdout, p.stderr)
