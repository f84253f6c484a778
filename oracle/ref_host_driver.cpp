// ref_host_driver.cpp -- what the reference's OWN host code produces for the driver's scene preparation (main.cu:59-71).
//
// TEST INFRASTRUCTURE (oracle/): nothing here is linked into, or run by, the product.
//
// Three of the reference's files are pure host C++ and compile here unmodified: happly.h (vendored PLY reader),
// matrix4x4.hpp and transform.hpp.  This driver -- own code -- performs the driver's calls on them and writes NUMBERS:
// the parsed vertex positions, the face index triples, the composite bunny matrix (main.cu:68-70), every vertex after
// Transform::apply and the Vec3 narrowing (main.cu:71,79-81), a set of Matrix4x4::Rotate matrices and of
// Transform::apply results.  tests/golden/make_ref_host_fixture.py turns the output into tests/golden/ref_host_fixture.npz;
// tests/test_host_api_cpp.py compiles THIS SAME FILE against the product's own headers (include/rtcuda/ply.hpp,
// matrix4x4.hpp, transform.hpp: -DREF_HOST_PRODUCT) and holds its output, and rtcuda_amd/scenes.py, to the fixture bit
// for bit.
//
//   reference side (oracle/Makefile, target _ref_host; only where /root/reference exists):
//       g++ -std=c++14 -O2 -ffp-contract=off -I/root/reference ref_host_driver.cpp -o _ref/ref_host
//   product side:
//       g++ -std=c++17 -O2 -ffp-contract=off -DREF_HOST_PRODUCT -I include oracle/ref_host_driver.cpp -o <tmp>/ref_host_product
//
//   ref_host <bun_zipper.ply> <out.bin>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef REF_HOST_PRODUCT
#define RTCUDA_PLY_AS_HAPPLY
#include "rtcuda/ply.hpp"
#include "rtcuda/matrix4x4.hpp"
#include "rtcuda/transform.hpp"
#else
#include "happly.h"         // /root/reference/happly.h       (by include path, unmodified)
#include "matrix4x4.hpp"    // /root/reference/matrix4x4.hpp
#include "transform.hpp"    // /root/reference/transform.hpp
#endif

static void put(FILE *f, const void *p, size_t n) {
    if (fwrite(p, 1, n, f) != n) throw std::runtime_error("short write");
}
static void put_matrix(FILE *f, const Matrix4x4 &m) {
    for (int i = 0; i < 4; i++) put(f, m.data[i], 16);
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s <mesh.ply> <out.bin>\n", argv[0]);
        return 2;
    }
    try {
        // main.cu:59-63
        happly::PLYData ply_in(argv[1]);
        std::vector<std::array<double, 3>> v_pos = ply_in.getVertexPositions();
        std::vector<std::vector<size_t>> f_index = ply_in.getFaceIndices<size_t>();
        FILE *f = fopen(argv[2], "wb");
        if (!f) return 2;
        const int64_t nv = (int64_t)v_pos.size(), nf = (int64_t)f_index.size();
        put(f, &nv, 8);
        put(f, &nf, 8);
        for (auto &v : v_pos) put(f, v.data(), 24);                       // parsed positions (double)
        for (auto &face : f_index) {                                      // face triples (int32)
            if (face.size() != 3) throw std::runtime_error("a face that is not a triangle");
            const int32_t t[3] = {(int32_t)face[0], (int32_t)face[1], (int32_t)face[2]};
            put(f, t, 12);
        }
        // main.cu:68-70
        Transform transform(Matrix4x4::Translate(0.0946899f, -0.0329874f, -0.0587997f));
        transform.composite(Matrix4x4::Scale(2.f, 2.f, 2.f));
        transform.composite(Matrix4x4::Translate(0.3f, 0.f, -0.5f));
        put_matrix(f, transform.matrix);
        // main.cu:71, then the narrowing of main.cu:79-81 (Vec3 holds floats); every 997th vertex also as apply() left it
        std::vector<std::array<double, 3>> kept;
        for (size_t i = 0; i < v_pos.size(); i++) {
            std::array<double, 3> v = v_pos[i];
            transform.apply(v);
            const float n3[3] = {(float)v[0], (float)v[1], (float)v[2]};
            put(f, n3, 12);
            if (i % 997 == 0) kept.push_back(v);
        }
        const int64_t nk = (int64_t)kept.size();
        put(f, &nk, 8);
        for (auto &v : kept) put(f, v.data(), 24);
        // Matrix4x4::Rotate (matrix4x4.hpp:36-56): unit axes, a diagonal, a non-unit axis, angles of both signs and > pi
        const float rot[12][4] = {{1.f, 0.f, 0.f, 0.5f},          {0.f, 1.f, 0.f, -1.25f},        {0.f, 0.f, 1.f, 3.0f},
                                  {0.577350259f, 0.577350259f, 0.577350259f, 0.7f}, {0.6f, 0.f, 0.8f, 2.5f}, {0.f, -0.6f, 0.8f, -0.1f},
                                  {0.267261237f, 0.534522474f, 0.801783681f, 1.0f}, {1.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 6.5f},
                                  {0.3f, 0.4f, 0.5f, 1.1f},        {-0.707106769f, 0.707106769f, 0.f, -3.1f}, {0.f, 0.f, -1.f, 1e-3f}};
        const int64_t nr = 12;
        put(f, &nr, 8);
        for (int k = 0; k < 12; k++) {
            put(f, rot[k], 16);
            put_matrix(f, Matrix4x4::Rotate(rot[k][0], rot[k][1], rot[k][2], rot[k][3]));
        }
        // Transform::composite + ::apply (transform.hpp:13-33) on the bunny transform followed by each rotation
        const double pts[4][3] = {{-0.0378297, 0.12794, 0.00447467}, {0.0, 0.0, 0.0}, {1.0, -2.0, 3.0}, {0.061, 0.1871, -0.0588}};
        const int64_t na = 12 * 4;
        put(f, &na, 8);
        for (int k = 0; k < 12; k++) {
            Transform t = transform;
            t.composite(Matrix4x4::Rotate(rot[k][0], rot[k][1], rot[k][2], rot[k][3]));
            for (int q = 0; q < 4; q++) {
                std::array<double, 3> p = {pts[q][0], pts[q][1], pts[q][2]};
                t.apply(p);
                put(f, p.data(), 24);
            }
        }
        fclose(f);
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_host: %s\n", e.what());
        return 1;
    }
    return 0;
}
