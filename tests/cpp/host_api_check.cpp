// host_api_check.cpp -- exercises the host side of the drop-in C++ API WITHOUT a GPU (tests/test_host_api_cpp.py
// builds and runs it): Vec3 arithmetic (vec3.cuh:32-147), Matrix4x4 incl. Rotate (matrix4x4.hpp:22-56), Transform
// (transform.hpp:13-33), the PLY reader (ASCII and binary) under happly's accessor names, the scene recipes
// (main.cu:41-166 + SURVEY 8d variants), point lights, and that Camera / render() at least compile and link.
//
//   host_api_check dump <variant> <ply> <out.bin>   : scene arrays as raw binary for comparison with scenes.py
//   host_api_check ply <file.ply> <out.bin>         : positions (double) + face indices (int64) as raw binary
//   host_api_check unit                             : prints name=value lines of small known-answer checks
//   host_api_check devices                          : RTCUDA_DEVICES as render() parses it (exit code 1: malformed)
#define RTCUDA_PLY_AS_HAPPLY
#include <cstdio>
#include <cstring>
#include <iostream>

#include "rtcuda/cornell_bunny.hpp"

static void put(FILE *f, const void *p, size_t n) {
    if (fwrite(p, 1, n, f) != n) throw std::runtime_error("short write");
}

static int dump(const std::string &variant, const std::string &ply, const std::string &out) {
    using rtcuda::CornellBunny;
    CornellBunny::Variant v = variant == "full_bsdf" ? CornellBunny::FULL_BSDF
                              : variant == "four_bunnies" ? CornellBunny::FOUR_BUNNIES
                              : variant == "sixteen_lights" ? CornellBunny::SIXTEEN_LIGHTS : CornellBunny::MATTE;
    CornellBunny cb(ply == "-" ? std::string() : ply, v);
    FILE *f = fopen(out.c_str(), "wb");
    if (!f) return 2;
    const int64_t n = (int64_t)cb.triangles().size(), nl = (int64_t)cb.lights().size();
    put(f, &n, 8);
    put(f, &nl, 8);
    for (const Triangle &t : cb.triangles()) {
        const float q[9] = {t.p0.x, t.p0.y, t.p0.z, t.p1_.x, t.p1_.y, t.p1_.z, t.p2_.x, t.p2_.y, t.p2_.z};
        put(f, q, sizeof(q));
    }
    for (int m : cb.material_of()) { int32_t x = m; put(f, &x, 4); }
    for (int l : cb.light_of()) { int32_t x = l; put(f, &x, 4); }
    for (const Light &l : cb.lights()) {
        int32_t tri = (int32_t)(l.d_triangle - &cb.triangles()[0]);
        const float L[3] = {l.L.x, l.L.y, l.L.z};
        put(f, &tri, 4);
        put(f, L, 12);
    }
    for (const Material &m : cb.materials()) {
        const float a[4] = {m.albedo.x, m.albedo.y, m.albedo.z, m.index_of_refraction};
        int32_t t = (int32_t)m.type;
        put(f, a, 16);
        put(f, &t, 4);
    }
    fclose(f);
    Scene sc = cb.scene();  // the Bvh / Scene aggregate the driver hands to render()
    return sc.bvh.num_primitives == (int)n && sc.num_lights == (int)nl ? 0 : 3;
}

static int ply(const std::string &path, const std::string &out) {
    happly::PLYData ply_in(path);  // the reference driver's three lines (main.cu:59-61)
    std::vector<std::array<double, 3>> v_pos = ply_in.getVertexPositions();
    std::vector<std::vector<size_t>> f_index = ply_in.getFaceIndices<size_t>();
    FILE *f = fopen(out.c_str(), "wb");
    if (!f) return 2;
    const int64_t nv = (int64_t)v_pos.size(), nf = (int64_t)f_index.size();
    put(f, &nv, 8);
    put(f, &nf, 8);
    for (auto &v : v_pos) put(f, v.data(), 24);
    for (auto &face : f_index) {
        int64_t k = (int64_t)face.size();
        put(f, &k, 8);
        for (size_t i : face) { int64_t x = (int64_t)i; put(f, &x, 8); }
    }
    fclose(f);
    return 0;
}

static void show(const char *name, const Vec3 &v) { printf("%s=%.9g %.9g %.9g\n", name, v.x, v.y, v.z); }
static void show(const char *name, float v) { printf("%s=%.9g\n", name, v); }

static int unit() {
    const Vec3 a(1.f, -2.f, 3.f), b(0.5f, 4.f, -0.25f);
    show("add", a + b);
    show("sub", a - b);
    show("mul", a * b);
    show("div", a / b);
    show("scale_l", 3.f * a);
    show("scale_r", a * 3.f);
    show("div_s", a / 3.f);  // multiplies by the fp32 reciprocal (vec3.cuh:56-59)
    show("div_s7", Vec3(5.f, 9.f, 13.f) / 7.f);  // (an input for which x * (1 / t) and x / t round differently)
    show("neg", -a);
    show("dot", dot(a, b));
    show("cross", cross(a, b));
    show("len", a.length());
    show("len2", a.length_squared());
    show("unit", a.unit_vector());
    show("max", a.max());
    show("reflect", reflect(Vec3(0.6f, -0.8f, 0.f), Vec3(0.f, 1.f, 0.f)));
    show("refract", refract(Vec3(0.6f, -0.8f, 0.f), Vec3(0.f, 1.f, 0.f), 1.0 / 1.5));
    Vec3 c = a;
    c += b; c -= Vec3(1.f); c *= Vec3(2.f, 3.f, 4.f); c /= Vec3(2.f); c *= 0.5f; c /= 3.f;
    show("compound", c);
    c = Vec3(4.f, 9.f, 2.f);
    c.sqrt_inplace();
    show("sqrt", c);
    show("zeros_ones", Vec3::make_zeros() + Vec3::make_ones());
    // Rotate about a unit axis; Transform::composite and ::apply
    const float s3 = 0.577350259f;
    Matrix4x4 r = Matrix4x4::Rotate(s3, s3, s3, 0.7f);
    for (int i = 0; i < 3; i++) printf("rot%d=%.9g %.9g %.9g %.9g\n", i, r.data[i][0], r.data[i][1], r.data[i][2], r.data[i][3]);
    Transform t(Matrix4x4::Translate(0.0946899f, -0.0329874f, -0.0587997f));
    t.composite(Matrix4x4::Scale(2.f, 2.f, 2.f));
    t.composite(Matrix4x4::Translate(0.3f, 0.f, -0.5f));
    for (int i = 0; i < 3; i++) printf("bunny%d=%.9g %.9g %.9g %.9g\n", i, t.matrix.data[i][0], t.matrix.data[i][1], t.matrix.data[i][2], t.matrix.data[i][3]);
    t.composite(r);
    std::array<double, 3> p = {-0.0378297, 0.12794, 0.00447467};
    t.apply(p);
    printf("applied=%.17g %.17g %.17g\n", p[0], p[1], p[2]);
    // a driver with a point light builds a Scene exactly as with area lights (light.cuh:70-76)
    std::vector<Material> mats = {Material::make_matte(Vec3(0.5f))};
    std::vector<Triangle> tris = {Triangle(Vec3(0, 0, 0), Vec3(1, 0, 0), Vec3(0, 1, 0))};
    std::vector<Light> lights = {Light::make_point_light(Vec3(0.5f, 0.5f, 1.f), Vec3(2.f))};
    std::vector<Primitive> prims = {Primitive(&tris[0], &mats[0])};
    Bvh bvh(tris, prims);
    Scene scene = {bvh, 1, lights.data()};
    printf("point_light=%d %d\n", (int)scene.d_lights[0].type, scene.bvh.num_primitives);
    // the reference's host PODs around a Triangle: members {p0, e1, e2, n}, accessors, BoundingBox, Intersection, Ray
    // (triangle.cuh:4-37, bounding_box.cuh:4-37, intersection.hpp:4-6, ray.cuh:4-25); the light triangle of SURVEY Appendix C
    Triangle lt(Vec3(0.4f, 0.999f, -0.4f), Vec3(0.6f, 0.999f, -0.4f), Vec3(0.6f, 0.999f, -0.6f));
    show("tri_e1", lt.e1); show("tri_e2", lt.e2); show("tri_n", lt.n);
    show("tri_p1", lt.p1()); show("tri_p2", lt.p2()); show("tri_center", lt.center());
    show("tri_puv", lt.p(0.25f, 0.5f));
    printf("tri_area=%.9g\n", lt.area());
    BoundingBox bb = lt.bounding_box();
    printf("tri_bbox=%.9g %.9g %.9g %.9g %.9g %.9g\n", bb.bounds[0], bb.bounds[1], bb.bounds[2], bb.bounds[3], bb.bounds[4], bb.bounds[5]);
    BoundingBox acc = BoundingBox::Empty();
    printf("bbox_empty=%d\n", acc.bounds[0] == FLT_MAX && acc.bounds[1] == -FLT_MAX && acc.bounds[5] == -FLT_MAX ? 1 : 0);
    acc.extend(bb);
    acc.extend(BoundingBox(0.f, 1.f, 0.f, 0.5f, -1.f, 0.f));
    printf("bbox_ext=%.9g %.9g %.9g %.9g %.9g %.9g\n", acc.bounds[0], acc.bounds[1], acc.bounds[2], acc.bounds[3], acc.bounds[4], acc.bounds[5]);
    printf("bbox_half_area=%.9g\n", acc.half_area());
    acc.reset();
    printf("bbox_reset=%d\n", acc.bounds[2] == FLT_MAX && acc.bounds[3] == -FLT_MAX ? 1 : 0);
    Intersection isect = {0.799000025f, 0.500000119f, 0.249999881f};
    Ray ray(Vec3(0.55f, 0.2f, -0.45f), Vec3(0.f, 1.f, 0.f));
    printf("ray_tmax_default=%d\n", ray.tmax == FLT_MAX ? 1 : 0);
    show("ray_at", ray.at(isect.t));
    show("offset_a", offset_ray_origin(Vec3(0.3f, 0.02f, -0.7f), Vec3(0.f, 1.f, 0.f)));
    Ray sp = Ray::spawn_offset_ray(Vec3(0.3f, 0.5f, -0.7f), Vec3(0.6f, -0.8f, 0.f), Vec3(0.f, 0.f, 1.f), 2.5f);
    show("offset_b", sp.origin);
    printf("spawn_tmax=%.9g\n", sp.tmax);
    return 0;
}

// the recipe's stage lines (profiler.hpp) on stdout, then one explicit stage and a scope guard
int stages(const std::string &ply) {
    rtcuda::CornellBunny cb(ply, rtcuda::CornellBunny::MATTE, true);
    profiler.start("Explicit stage");
    profiler.stop();
    {
        Profiler::Stage s(profiler, "Scoped stage");
    }
    profiler.enabled = false;
    profiler.start("Silent stage");
    profiler.stop();
    printf("triangles=%zu last_ms_ok=%d\n", cb.triangles().size(), profiler.last_ms >= 0.f ? 1 : 0);
    return 0;
}

// never called without a GPU: proves that the render path of the API compiles and links against the library
int render_smoke(const std::string &ply) {
    rtcuda::CornellBunny cb(ply);
    std::vector<Vec3> framebuffer;
    render(64, 36, 4, 10, rtcuda::CornellBunny::camera(64.f / 36.f), cb.scene(), framebuffer);
    return (int)framebuffer.size();
}

int main(int argc, char **argv) {
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "dump" && argc == 5) return dump(argv[2], argv[3], argv[4]);
        if (mode == "ply" && argc == 4) return ply(argv[2], argv[3]);
        if (mode == "unit") return unit();
        if (mode == "devices") {  // RTCUDA_DEVICES as render() reads it: the list, or an exception (exit code 1) on a malformed value
            for (int d : devices_from_env()) printf("%d ", d);
            printf("\n");
            return 0;
        }
        if (mode == "stages" && argc == 3) return stages(argv[2]);
        if (mode == "render" && argc == 3) return render_smoke(argv[2]) > 0 ? 0 : 1;
        fprintf(stderr, "usage: host_api_check dump <variant> <ply|-> <out> | ply <in> <out> | unit | render <ply>\n");
        return 64;
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
