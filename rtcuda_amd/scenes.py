"""Scene recipes for the render path, as flat numpy arrays (the shape the C-ABI takes).

This is the caller side of the hot path: it restates what the reference's driver does before it
calls ``render()`` -- load ``bun_zipper.ply``, transform it, append the Cornell box and the two
light triangles (``main.cu:41-166``) -- so that the HIP library and the CPU oracle are fed the
SAME bytes.  Everything here is host-side numpy; no GPU work.

Reference semantics reproduced on purpose:

* PLY ASCII floats are parsed to fp32 and then promoted to double
  (``happly.h:318-325``: ``istringstream >> float``; ``getVertexPositions`` returns doubles).
* ``Transform::composite`` accumulates ``other . matrix`` in fp32 (``transform.hpp:13-24``).
* ``Transform::apply`` rounds x and y to fp32 but keeps z in double until the ``Vec3`` narrowing
  (``transform.hpp:26-33``).
* Light order is the reference's ``std::unordered_map`` iteration order (``main.cu:110-131``):
  light 0 is the LAST light triangle (index 69462), light 1 the one before it (SURVEY Appx A.13).

Variant scenes (C2 full-BSDF, C4 four bunnies, C5 sixteen lights) are this project's definitions
from SURVEY.md section 8d -- the reference ships only the all-matte scene.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

MATTE, MIRROR, GLASS = 0, 1, 2
POINT_LIGHT, AREA_LIGHT = 0, 1

MATERIAL_DTYPE = np.dtype([("albedo", np.float32, 3), ("ior", np.float32), ("type", np.int32)])
LIGHT_DTYPE = np.dtype([("type", np.int32), ("pos", np.float32, 3), ("tri", np.int32), ("L", np.float32, 3)])

_DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
BUNNY_PLY = os.path.join(_DATA_DIR, "bun_zipper.ply")


@dataclass
class SceneArrays:
    """Flat scene description: what ``rt_scene_create`` / ``orc_scene_create`` consume."""

    tris: np.ndarray          # (n, 9) float32: p0 p1 p2
    tri_material: np.ndarray  # (n,) int32
    tri_light: np.ndarray     # (n,) int32, -1 = not a light
    materials: np.ndarray     # MATERIAL_DTYPE
    lights: np.ndarray        # LIGHT_DTYPE
    name: str = ""
    meta: dict = field(default_factory=dict)

    @property
    def n_tris(self) -> int:
        return int(self.tris.shape[0])


@dataclass
class CameraParams:
    lookfrom: tuple = (0.5, 0.5, 1.5)
    lookat: tuple = (0.5, 0.5, 0.0)
    up: tuple = (0.0, 1.0, 0.0)
    vfov: float = 37.8


# --------------------------------------------------------------------------- PLY ingest
def load_ply(path: str = BUNNY_PLY):
    """Minimal PLY reader (ASCII and binary_little_endian): vertex x,y,z and triangle faces.

    Returns (positions float32 (nv, 3), faces int64 (nf, 3)).  ASCII floats go through fp32,
    exactly as ``happly``'s ``istringstream >> float`` does for a ``property float``.
    """
    with open(path, "rb") as fh:
        header = []
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("PLY: no end_header")
            header.append(line.decode("ascii", "replace").strip())
            if header[-1] == "end_header":
                break
        body = fh.read()
    if header[0] != "ply":
        raise ValueError("PLY: bad magic")
    fmt = None
    elements = []  # (name, count, [(kind, name, types...)])
    for ln in header[1:]:
        tok = ln.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append([tok[1], int(tok[2]), []])
        elif tok[0] == "property":
            if tok[1] == "list":
                elements[-1][2].append(("list", tok[4], tok[2], tok[3]))
            else:
                elements[-1][2].append(("scalar", tok[2], tok[1]))
    np_types = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
                "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2",
                "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}
    pos = faces = None
    if fmt == "ascii":
        tokens = body.split()
        cur = 0
        for name, count, props in elements:
            if name == "vertex":
                ncol = len(props)
                block = tokens[cur:cur + count * ncol]
                cur += count * ncol
                arr = np.array(block, dtype="S32").reshape(count, ncol)
                names = [p[1] for p in props]
                cols = [names.index(a) for a in ("x", "y", "z")]
                # decimal text -> fp32 (the double detour cannot change the rounding of these
                # <= 9-significant-digit literals)
                pos = np.stack([arr[:, c].astype(np.float64).astype(np.float32) for c in cols], axis=1)
            elif name == "face":
                fl = []
                for _ in range(count):
                    k = int(tokens[cur])
                    fl.append([int(t) for t in tokens[cur + 1:cur + 1 + k]])
                    cur += 1 + k
                if any(len(f) != 3 for f in fl):
                    raise ValueError("PLY: only triangle faces are supported")
                faces = np.array(fl, dtype=np.int64)
            else:
                for _ in range(count):
                    for p in props:
                        if p[0] == "list":
                            k = int(tokens[cur])
                            cur += 1 + k
                        else:
                            cur += 1
    elif fmt == "binary_little_endian":
        off = 0
        for name, count, props in elements:
            if all(p[0] == "scalar" for p in props):
                dt = np.dtype([(p[1], "<" + np_types[p[2]]) for p in props])
                rec = np.frombuffer(body, dtype=dt, count=count, offset=off)
                off += dt.itemsize * count
                if name == "vertex":
                    pos = np.stack([rec[a].astype(np.float32) for a in ("x", "y", "z")], axis=1)
            else:
                fl = []
                for _ in range(count):
                    row = None
                    for p in props:
                        if p[0] == "list":
                            ct = np.dtype("<" + np_types[p[2]])
                            it = np.dtype("<" + np_types[p[3]])
                            k = int(np.frombuffer(body, ct, 1, off)[0])
                            off += ct.itemsize
                            vals = np.frombuffer(body, it, k, off)
                            off += it.itemsize * k
                            if p[1] in ("vertex_indices", "vertex_index"):
                                row = [int(v) for v in vals]
                        else:
                            off += np.dtype(np_types[p[2]]).itemsize
                    if name == "face":
                        if row is None or len(row) != 3:
                            raise ValueError("PLY: only triangle faces are supported")
                        fl.append(row)
                if name == "face":
                    faces = np.array(fl, dtype=np.int64)
    else:
        raise ValueError(f"PLY: unsupported format {fmt!r}")
    if pos is None or faces is None:
        raise ValueError("PLY: missing vertex or face element")
    return pos, faces


# --------------------------------------------------------------------------- transforms
def translate(dx, dy, dz):
    """``Matrix4x4::Translate`` (matrix4x4.hpp:22-27)."""
    m = np.eye(4, dtype=np.float32)
    m[0, 3], m[1, 3], m[2, 3] = np.float32(dx), np.float32(dy), np.float32(dz)
    return m


def scale(sx, sy, sz):
    """``Matrix4x4::Scale`` (matrix4x4.hpp:29-34)."""
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2] = np.float32(sx), np.float32(sy), np.float32(sz)
    return m


def rotate(ax, ay, az, theta_rad):
    """``Matrix4x4::Rotate`` (matrix4x4.hpp:36-56), fp32 throughout (cosf/sinf = numpy fp32)."""
    x, y, z = np.float32(ax), np.float32(ay), np.float32(az)
    c = np.cos(np.float32(theta_rad), dtype=np.float32)
    s = np.sin(np.float32(theta_rad), dtype=np.float32)
    c1 = np.float32(1.0) - c
    m = np.eye(4, dtype=np.float32)
    m[0, :3] = [c + x * x * c1, x * y * c1 - z * s, x * z * c1 + y * s]
    m[1, :3] = [x * y * c1 + z * s, c + y * y * c1, y * z * c1 - x * s]
    m[2, :3] = [x * z * c1 - y * s, y * z * c1 + x * s, c + z * z * c1]
    return m


def composite(matrix, other):
    """``Transform::composite`` (transform.hpp:13-24): result = other . matrix, fp32, k ascending."""
    res = np.zeros((4, 4), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            acc = np.float32(0.0)
            for k in range(4):
                acc = np.float32(acc + np.float32(other[i, k] * matrix[k, j]))
            res[i, j] = acc
    return res


def apply_transform(matrix, v_pos_f64):
    """``Transform::apply`` (transform.hpp:26-33) followed by the ``Vec3`` narrowing (main.cu:79-81).

    x and y are computed in double and rounded to fp32; z is computed in double from the ORIGINAL
    x, y and only narrows to fp32 when the triangle is built.  Returns float32 (n, 3).
    """
    m = matrix.astype(np.float64)
    v0, v1, v2 = v_pos_f64[:, 0], v_pos_f64[:, 1], v_pos_f64[:, 2]
    nx = (((m[0, 0] * v0 + m[0, 1] * v1) + m[0, 2] * v2) + m[0, 3]).astype(np.float32)
    ny = (((m[1, 0] * v0 + m[1, 1] * v1) + m[1, 2] * v2) + m[1, 3]).astype(np.float32)
    nz = (((m[2, 0] * v0 + m[2, 1] * v1) + m[2, 2] * v2) + m[2, 3]).astype(np.float32)
    return np.stack([nx, ny, nz], axis=1)


def bunny_transform(extra_translate=(0.3, 0.0, -0.5)):
    """main.cu:68-70: Translate(T1) then composite(Scale 2) then composite(Translate T2)."""
    m = translate(0.0946899, -0.0329874, -0.0587997)
    m = composite(m, scale(2.0, 2.0, 2.0))
    m = composite(m, translate(*extra_translate))
    return m


# --------------------------------------------------------------------------- Cornell box
_WALLS = [  # main.cu:88-107 ; material index per triangle: 0 red, 1 green, 2 white
    ((0, 0, 0), (0, 0, -1), (0, 1, -1), 0),
    ((0, 0, 0), (0, 1, 0), (0, 1, -1), 0),
    ((1, 0, 0), (1, 0, -1), (1, 1, -1), 1),
    ((1, 0, 0), (1, 1, 0), (1, 1, -1), 1),
    ((0, 0, 0), (1, 0, 0), (1, 0, -1), 2),
    ((0, 0, 0), (0, 0, -1), (1, 0, -1), 2),
    ((0, 1, 0), (1, 1, 0), (1, 1, -1), 2),
    ((0, 1, 0), (0, 1, -1), (1, 1, -1), 2),
    ((0, 0, -1), (1, 0, -1), (1, 1, -1), 2),
    ((0, 0, -1), (0, 1, -1), (1, 1, -1), 2),
]
_REF_LIGHTS = [  # main.cu:111-116
    ((0.4, 0.999, -0.4), (0.6, 0.999, -0.4), (0.6, 0.999, -0.6)),
    ((0.4, 0.999, -0.4), (0.4, 0.999, -0.6), (0.6, 0.999, -0.6)),
]


def _materials(bunny_glass=False, back_mirror=False):
    mats = np.zeros(6, dtype=MATERIAL_DTYPE)
    mats[0] = ((0.65, 0.05, 0.05), 0.0, MATTE)   # red    main.cu:42
    mats[1] = ((0.12, 0.45, 0.15), 0.0, MATTE)   # green  :43
    mats[2] = ((0.73, 0.73, 0.73), 0.0, MATTE)   # white  :44
    mats[3] = ((0.62, 0.57, 0.54), 0.0, MATTE)   # brown  :45
    mats[4] = ((0.0, 0.0, 0.0), 1.5, GLASS)      # C2: glass bunny
    mats[5] = ((0.9, 0.9, 0.9), 0.0, MIRROR)     # C2: mirror back wall
    return mats


def cornell_bunny(variant: str = "matte", ply_path: str = BUNNY_PLY, bunny: bool = True) -> SceneArrays:
    """Build one of the benchmark scenes.

    variant: "matte" (the reference scene, C1/C3), "full_bsdf" (C2: glass bunny + mirror back wall),
    "four_bunnies" (C4), "sixteen_lights" (C5).  ``bunny=False`` gives the bare box (tiny tests).
    """
    if variant not in ("matte", "full_bsdf", "four_bunnies", "sixteen_lights"):
        raise ValueError(f"unknown scene variant {variant!r}")
    tri_list, mat_list = [], []
    if bunny:
        pos, faces = load_ply(ply_path)
        v64 = pos.astype(np.float64)
        if variant == "four_bunnies":
            places = [(0.1, 0.0, -0.3), (0.55, 0.0, -0.3), (0.1, 0.0, -0.62), (0.55, 0.0, -0.62)]
        else:
            places = [(0.3, 0.0, -0.5)]
        bunny_mat = 4 if variant == "full_bsdf" else 3
        for place in places:
            vt = apply_transform(bunny_transform(place), v64)
            tri_list.append(vt[faces].reshape(-1, 9))
            mat_list.append(np.full(faces.shape[0], bunny_mat, dtype=np.int32))
    walls = np.array([[*a, *b, *c] for a, b, c, _ in _WALLS], dtype=np.float32)
    wall_mats = np.array([m for *_, m in _WALLS], dtype=np.int32)
    if variant == "full_bsdf":
        wall_mats[8] = 5
        wall_mats[9] = 5
    tri_list.append(walls)
    mat_list.append(wall_mats)
    if variant == "sixteen_lights":
        quads = []
        h = np.float32(0.05)  # corners are computed in fp32 (matches SURVEY Appendix C's numbers)
        for cz in (-0.35, -0.65):
            for cx in (0.2, 0.4, 0.6, 0.8):
                cx, cz = np.float32(cx), np.float32(cz)
                x0, x1, z0, z1 = cx - h, cx + h, cz + h, cz - h
                quads.append(((x0, 0.999, z0), (x1, 0.999, z0), (x1, 0.999, z1)))
                quads.append(((x0, 0.999, z0), (x0, 0.999, z1), (x1, 0.999, z1)))
        light_tris = np.array([[*a, *b, *c] for a, b, c in quads], dtype=np.float32)
        radiance = 7.5
    else:
        light_tris = np.array([[*a, *b, *c] for a, b, c in _REF_LIGHTS], dtype=np.float32)
        radiance = 15.0
    n_before = sum(t.shape[0] for t in tri_list)
    tri_list.append(light_tris)
    mat_list.append(np.full(light_tris.shape[0], 2, dtype=np.int32))  # lights are white matte too
    tris = np.ascontiguousarray(np.concatenate(tri_list, axis=0), dtype=np.float32)
    tri_material = np.ascontiguousarray(np.concatenate(mat_list), dtype=np.int32)
    n_l = light_tris.shape[0]
    light_tri_idx = np.arange(n_before, n_before + n_l, dtype=np.int32)
    if variant != "sixteen_lights":
        light_tri_idx = light_tri_idx[::-1].copy()  # unordered_map order: [69462, 69461]
    lights = np.zeros(n_l, dtype=LIGHT_DTYPE)
    tri_light = np.full(tris.shape[0], -1, dtype=np.int32)
    for k, ti in enumerate(light_tri_idx):
        lights[k] = (AREA_LIGHT, (0, 0, 0), int(ti), (radiance,) * 3)
        tri_light[ti] = k
    return SceneArrays(tris=tris, tri_material=tri_material, tri_light=tri_light, materials=_materials(),
                       lights=lights, name=variant + ("" if bunny else "_box"),
                       meta={"variant": variant, "bunny": bunny})


def write_ppm(path: str, image: np.ndarray) -> None:
    """The driver's tone step (main.cu:178-191): ``clamp(int(256*c), 0, 255)``, P3 text."""
    h, w, _ = image.shape
    q = np.clip((np.float32(256.0) * image.astype(np.float32)).astype(np.int64), 0, 255)
    with open(path, "w") as fh:
        fh.write(f"P3\n{w} {h}\n255\n")
        for row in q.reshape(-1, 3):
            fh.write(f"{row[0]} {row[1]} {row[2]}\n")
