// rtcuda.hpp -- the reference's host API for the render path, on top of the C-ABI (rtcuda_amd.h).
//
// Same names, constructor signatures and call sequence as lashhw/rtcuda's headers, so its driver
// (main.cu:41-173) compiles against this file with the CUDA allocation calls deleted:
//
//   Vec3 + operators, dot, cross, ...     vec3.cuh:4-147       (everything but the device-only atomic_add)
//   Matrix4x4, Transform                  matrix4x4.hpp, transform.hpp   (included below: rtcuda/matrix4x4.hpp, transform.hpp)
//   PLY ingest                            happly.h (vendored)  -> rtcuda/ply.hpp (own reader, happly's two accessor names)
//   Triangle(p0, p1, p2) + p1() p2() center() bounding_box() p(u, v), members p0 e1 e2 n    triangle.cuh:4-37
//   BoundingBox, Intersection, Ray        bounding_box.cuh:4-37, intersection.hpp:4-6, ray.cuh:4-25
//   Material::make_matte/mirror/glass     material.cuh:25-44
//   Light::make_point_light/area_light    light.cuh:70-84
//   Primitive(tri*, mat*, light* = NULL)  primitive.cuh:6-7
//   Bvh(triangles, primitives)            bvh.cuh:17,30        (asserts equal sizes, :33)
//   Scene{bvh, num_lights, d_lights}      scene.cuh:4-8
//   Camera(lookfrom, lookat, up, vfov, aspect)  camera.cuh:6,15
//   render(w, h, spp, max_bounces, camera, scene, framebuffer)  render.cuh:366-367
//
// Difference by design: the pointers handed to Light / Primitive are HOST pointers into the
// caller's own arrays (the reference hands out device pointers it cudaMalloc'ed and never frees);
// render() flattens them to indices and the library owns every device allocation.
// Errors throw std::runtime_error (the reference prints and exit()s, utility.cuh:6-13).
#ifndef RTCUDA_HPP
#define RTCUDA_HPP

#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <cfloat>
#include <stdexcept>
#include <string>
#include <vector>

#include "../rtcuda_amd.h"
#include "matrix4x4.hpp"
#include "ply.hpp"
#include "profiler.hpp"
#include "transform.hpp"

// vec3.cuh:4-147, host side.  Arithmetic forms are the reference's where they round differently from the obvious
// spelling: division by a scalar multiplies by the fp32 reciprocal (vec3.cuh:56-59, :123-129), unit_vector multiplies
// by 1 / length (:131-134), Vec3 / Vec3 is a true per-component divide (:44-46), refract takes its ratio as double.
struct Vec3 {
    Vec3() {}
    constexpr Vec3(float x, float y, float z) : x(x), y(y), z(z) {}
    constexpr Vec3(float xyz) : x(xyz), y(xyz), z(xyz) {}
    static Vec3 make_zeros() { return Vec3(0.f, 0.f, 0.f); }
    static Vec3 make_ones() { return Vec3(1.f, 1.f, 1.f); }

    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    Vec3 &operator+=(const Vec3 &b) { x += b.x; y += b.y; z += b.z; return *this; }
    Vec3 &operator-=(const Vec3 &b) { x -= b.x; y -= b.y; z -= b.z; return *this; }
    Vec3 &operator*=(const Vec3 &b) { x *= b.x; y *= b.y; z *= b.z; return *this; }
    Vec3 &operator/=(const Vec3 &b) { x /= b.x; y /= b.y; z /= b.z; return *this; }
    Vec3 &operator*=(float t) { x *= t; y *= t; z *= t; return *this; }
    Vec3 &operator/=(float t) { return *this *= 1.f / t; }

    float max() const { return fmaxf(fmaxf(x, y), z); }
    float length_squared() const { return x * x + y * y + z * z; }
    float length() const { return sqrtf(length_squared()); }
    Vec3 unit_vector() const { Vec3 r = *this; r.unit_vector_inplace(); return r; }
    void unit_vector_inplace() { *this *= 1.f / length(); }
    void sqrt_inplace() { x = sqrtf(x); y = sqrtf(y); z = sqrtf(z); }

    float x, y, z;
};
inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(const Vec3 &a, const Vec3 &b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3 operator/(const Vec3 &a, const Vec3 &b) { return Vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline Vec3 operator*(const Vec3 &v, float t) { return Vec3(v.x * t, v.y * t, v.z * t); }
inline Vec3 operator*(float t, const Vec3 &v) { return v * t; }
inline Vec3 operator/(const Vec3 &v, float t) { return v * (1.f / t); }
inline float dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Vec3 reflect(const Vec3 &v, const Vec3 &unit_n) { return v - 2.f * dot(v, unit_n) * unit_n; }
// vec3.cuh:82-86 (the double ratio narrows to float at the scalar * Vec3 product); :76-80 derives cos_theta itself
inline Vec3 refract(const Vec3 &unit_v, const Vec3 &unit_n, double eta_ratio, float cos_theta) {
    const Vec3 parallel = (float)eta_ratio * (unit_v + cos_theta * unit_n);
    return parallel + -sqrtf(1.f - parallel.length_squared()) * unit_n;
}
inline Vec3 refract(const Vec3 &unit_v, const Vec3 &unit_n, double eta_ratio) {
    return refract(unit_v, unit_n, eta_ratio, (float)(double)-dot(unit_v, unit_n));
}

// bounding_box.cuh:4-37 -- bounds = [xmin, xmax, ymin, ymax, zmin, zmax]
struct BoundingBox {
    BoundingBox() {}
    BoundingBox(float xmin, float xmax, float ymin, float ymax, float zmin, float zmax) {
        const float b[6] = {xmin, xmax, ymin, ymax, zmin, zmax};
        for (int k = 0; k < 6; k++) bounds[k] = b[k];
    }
    static BoundingBox Empty() { BoundingBox b; b.reset(); return b; }
    void reset() {
        for (int a = 0; a < 3; a++) {
            bounds[2 * a] = FLT_MAX;
            bounds[2 * a + 1] = -FLT_MAX;
        }
    }
    void extend(const BoundingBox &other) {
        for (int a = 0; a < 3; a++) {
            bounds[2 * a] = fminf(bounds[2 * a], other.bounds[2 * a]);
            bounds[2 * a + 1] = fmaxf(bounds[2 * a + 1], other.bounds[2 * a + 1]);
        }
    }
    float half_area() const {  // (dx + dy) * dz + dx * dy, in that order (bounding_box.cuh:27-32)
        const float dx = bounds[1] - bounds[0], dy = bounds[3] - bounds[2], dz = bounds[5] - bounds[4];
        return (dx + dy) * dz + dx * dy;
    }
    float bounds[6];
};

// intersection.hpp:4-6
struct Intersection {
    float t, u, v;
};

// utility.cuh:31-47 on the host: the self-intersection offset of a spawned ray's origin (integer steps on the bit
// pattern, or n / 65536 near the coordinate planes).  The device code (rt_device.h) is what renders; this is the same
// function for drivers that spawn rays themselves.
inline Vec3 offset_ray_origin(const Vec3 &p, const Vec3 &n) {
    auto one = [](float pc, float nc) {
        if (fabsf(pc) < 1.f / 32.f) return pc + (1.f / 65536.f) * nc;
        int32_t bits, step = (int32_t)(256.f * nc);
        memcpy(&bits, &pc, 4);
        bits += pc < 0 ? -step : step;
        float r;
        memcpy(&r, &bits, 4);
        return r;
    };
    return Vec3(one(p.x, n.x), one(p.y, n.y), one(p.z, n.z));
}

// ray.cuh:4-25
struct Ray {
    Ray() {}
    Ray(const Vec3 &origin, const Vec3 &unit_d, float tmax = FLT_MAX) : origin(origin), unit_d(unit_d), tmax(tmax) {}
    Vec3 at(float t) const { return origin + t * unit_d; }
    static Ray spawn_offset_ray(const Vec3 &origin, const Vec3 &unit_n, const Vec3 &unit_d, float tmax = FLT_MAX) {
        return Ray(offset_ray_origin(origin, unit_n), unit_d, tmax);
    }
    Vec3 origin, unit_d;
    float tmax;
};

// triangle.cuh:4-37, host side: the reference's public members {p0, e1 = p0 - p1, e2 = p2 - p0, n = e1 x e2} and its
// accessors.  Note p1() = p0 - e1 and p2() = p0 + e2 are the reference's ROUNDED reconstructions, not the constructor's
// arguments; those are kept beside them (p1_, p2_) because the library derives its own e1, e2, n from the three
// vertices with pinned fp32 arithmetic (rt_scene_create), exactly as the reference's constructor does on the host.
struct Triangle {
    Triangle() {}
    Triangle(const Vec3 &p0, const Vec3 &p1, const Vec3 &p2)
        : p0(p0), e1(p0 - p1), e2(p2 - p0), n(cross(e1, e2)), p1_(p1), p2_(p2) {}
    Vec3 p1() const { return p0 - e1; }
    Vec3 p2() const { return p0 + e2; }
    Vec3 center() const { return (p0 + p1() + p2()) * (1.f / 3.f); }
    Vec3 p(float u, float v) const { return p0 - u * e1 + v * e2; }  // triangle.cuh:15
    float area() const { return 0.5f * n.length(); }                 // triangle.cuh:84-86
    BoundingBox bounding_box() const {
        const Vec3 a = p1(), b = p2();
        return BoundingBox(fminf(p0.x, fminf(a.x, b.x)), fmaxf(p0.x, fmaxf(a.x, b.x)), fminf(p0.y, fminf(a.y, b.y)),
                           fmaxf(p0.y, fmaxf(a.y, b.y)), fminf(p0.z, fminf(a.z, b.z)), fmaxf(p0.z, fmaxf(a.z, b.z)));
    }
    Vec3 p0, e1, e2, n;
    Vec3 p1_, p2_;  // the constructor's second and third vertex, as given
};

enum MaterialType { MATTE, MIRROR, GLASS };
struct Material {
    Material() : albedo(0.f), index_of_refraction(0.f), type(MATTE) {}
    static Material make_matte(const Vec3 &albedo) { Material m; m.albedo = albedo; m.type = MATTE; return m; }
    static Material make_mirror(const Vec3 &albedo) { Material m; m.albedo = albedo; m.type = MIRROR; return m; }
    static Material make_glass(float ior) { Material m; m.index_of_refraction = ior; m.type = GLASS; return m; }
    Vec3 albedo;
    float index_of_refraction;
    MaterialType type;
};

enum LightType { POINT_LIGHT, AREA_LIGHT };
struct Light {
    Light() : type(POINT_LIGHT), pos(0.f), d_triangle(nullptr), L(0.f) {}
    static Light make_point_light(const Vec3 &pos, const Vec3 &I) { Light l; l.type = POINT_LIGHT; l.pos = pos; l.L = I; return l; }
    static Light make_area_light(Triangle *d_triangle, const Vec3 &L) { Light l; l.type = AREA_LIGHT; l.d_triangle = d_triangle; l.L = L; return l; }
    LightType type;
    Vec3 pos;
    Triangle *d_triangle;
    Vec3 L;  // I for point lights
};

struct Primitive {
    Primitive() : d_triangle(nullptr), d_mat(nullptr), d_area_light(nullptr) {}
    Primitive(Triangle *d_triangle, Material *d_mat, Light *d_area_light = nullptr)
        : d_triangle(d_triangle), d_mat(d_mat), d_area_light(d_area_light) {}
    Triangle *d_triangle;
    Material *d_mat;
    Light *d_area_light;
};

namespace rtcuda_detail {
struct SceneHandle {
    rt_scene *h = nullptr;
    const Light *lights_key = nullptr;
    int num_lights_key = -1;
    ~SceneHandle() { if (h) rt_scene_destroy(h); }
};
inline void check(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}
}  // namespace rtcuda_detail

// Host-side description only; the device BVH is built by the library when render() first sees the scene.
struct Bvh {
    Bvh() : num_primitives(0) {}
    Bvh(const std::vector<Triangle> &triangles, const std::vector<Primitive> &primitives)
        : num_primitives((int)triangles.size()), triangles(triangles), primitives(primitives),
          triangle_base(triangles.empty() ? nullptr : &triangles[0]),
          handle(std::make_shared<rtcuda_detail::SceneHandle>()) {
        assert(triangles.size() == primitives.size());  // bvh.cuh:33
    }
    int num_primitives;
    std::vector<Triangle> triangles;
    std::vector<Primitive> primitives;
    const Triangle *triangle_base;  // primitives[i].d_triangle and Light::d_triangle point into [base, base + n)
    std::shared_ptr<rtcuda_detail::SceneHandle> handle;
};

struct Scene {
    Bvh bvh;
    int num_lights;
    Light *d_lights;
};

struct Camera {
    Camera() {}
    Camera(Vec3 lookfrom, Vec3 lookat, Vec3 up, float vfov, float aspect_ratio) {
        float a[3] = {lookfrom.x, lookfrom.y, lookfrom.z}, b[3] = {lookat.x, lookat.y, lookat.z}, c[3] = {up.x, up.y, up.z};
        rtcuda_detail::check(rt_camera_make(a, b, c, vfov, aspect_ratio, &pod), "Camera");
    }
    rt_camera pod;
};

namespace rtcuda_detail {
inline rt_scene *realise(const Scene &scene) {
    SceneHandle &sh = *scene.bvh.handle;
    if (sh.h && sh.lights_key == scene.d_lights && sh.num_lights_key == scene.num_lights) return sh.h;
    if (sh.h) { rt_scene_destroy(sh.h); sh.h = nullptr; }
    const Bvh &bvh = scene.bvh;
    const int n = bvh.num_primitives;
    std::vector<float> verts((size_t)9 * (n > 0 ? n : 1));
    std::vector<int32_t> tri_mat(n > 0 ? n : 1), tri_light(n > 0 ? n : 1);
    std::vector<const Material *> mat_ptrs;
    std::vector<rt_material> mats;
    // primitives may come in any order: primitive i describes triangle (d_triangle - base)
    for (int i = 0; i < n; i++) {
        const Primitive &pr = bvh.primitives[i];
        long ti = pr.d_triangle - bvh.triangle_base;
        if (ti < 0 || ti >= n) throw std::runtime_error("render: Primitive::d_triangle does not point into the Bvh's triangles");
        const Triangle &t = bvh.triangles[ti];
        float *q = &verts[9 * (size_t)ti];
        q[0] = t.p0.x; q[1] = t.p0.y; q[2] = t.p0.z;
        q[3] = t.p1_.x; q[4] = t.p1_.y; q[5] = t.p1_.z;
        q[6] = t.p2_.x; q[7] = t.p2_.y; q[8] = t.p2_.z;
        int mi = -1;
        for (size_t k = 0; k < mat_ptrs.size(); k++) if (mat_ptrs[k] == pr.d_mat) { mi = (int)k; break; }
        if (mi < 0) {
            if (!pr.d_mat) throw std::runtime_error("render: Primitive without material");
            mi = (int)mat_ptrs.size();
            mat_ptrs.push_back(pr.d_mat);
            rt_material m;
            m.albedo[0] = pr.d_mat->albedo.x; m.albedo[1] = pr.d_mat->albedo.y; m.albedo[2] = pr.d_mat->albedo.z;
            m.index_of_refraction = pr.d_mat->index_of_refraction;
            m.type = (int32_t)pr.d_mat->type;
            mats.push_back(m);
        }
        tri_mat[ti] = mi;
        long li = pr.d_area_light ? pr.d_area_light - scene.d_lights : -1;
        if (pr.d_area_light && (li < 0 || li >= scene.num_lights)) throw std::runtime_error("render: Primitive::d_area_light does not point into Scene::d_lights");
        tri_light[ti] = (int32_t)li;
    }
    std::vector<rt_light> lights(scene.num_lights > 0 ? scene.num_lights : 1);
    for (int k = 0; k < scene.num_lights; k++) {
        const Light &l = scene.d_lights[k];
        rt_light &o = lights[k];
        o.type = (int32_t)l.type;
        o.pos[0] = l.pos.x; o.pos[1] = l.pos.y; o.pos[2] = l.pos.z;
        o.L[0] = l.L.x; o.L[1] = l.L.y; o.L[2] = l.L.z;
        o.triangle = -1;
        if (l.type == AREA_LIGHT) {
            long ti = l.d_triangle - bvh.triangle_base;
            if (ti < 0 || ti >= n) throw std::runtime_error("render: Light::d_triangle does not point into the Bvh's triangles");
            o.triangle = (int32_t)ti;
        }
    }
    check(rt_scene_create(verts.data(), n, tri_mat.data(), tri_light.data(), mats.data(), (int)mats.size(),
                          lights.data(), scene.num_lights, &sh.h), "rt_scene_create");
    sh.lights_key = scene.d_lights;
    sh.num_lights_key = scene.num_lights;
    return sh.h;
}
}  // namespace rtcuda_detail

// The reference builds and uploads its BVH in the Bvh constructor (bvh.cuh:30-219); here the device scene is created
// the first time a Scene is rendered.  prepare() does it ahead of time, so a driver can time the two apart.
inline void prepare(const Scene &scene) { (void)rtcuda_detail::realise(scene); }

// A driver that must keep the reference's exact call (main.cu:173) can still reach several GPUs: RTCUDA_DEVICES="0,1,2,3" in
// the environment sends the seven-argument render() below through the multi-device path (rt_render_multi).
inline std::vector<int> devices_from_env() {
    std::vector<int> d;
    if (const char *e = std::getenv("RTCUDA_DEVICES")) {
        // a typo must not silently change which GPUs render: anything but non-negative integers separated by
        // single commas is an error, and so is an empty list
        const std::string text(e);
        const char *q = e;
        while (true) {
            char *end = nullptr;
            const long v = std::strtol(q, &end, 10);
            if (end == q || v < 0 || v > 1 << 20 || (*end != ',' && *end != 0))
                throw std::runtime_error("RTCUDA_DEVICES=\"" + text + "\": expected a comma-separated list of device ordinals, e.g. 0,1,2,3");
            d.push_back((int)v);
            if (*end == 0) break;
            q = end + 1;
        }
    }
    return d;
}

// ... and the library's modes: RTCUDA_WATERTIGHT=1 (RT_FLAG_WATERTIGHT: the triangle-list definition of the hits instead of the
// reference's -- no accepted hit lost to a box test, ties by caller index), RTCUDA_REFERENCE_WALK=1 (RT_FLAG_REFERENCE_WALK: every
// ray through the reference's own tree; the same image as the default, several times slower: a cross-check) and
// RTCUDA_DETERMINISTIC=1 (RT_FLAG_DETERMINISTIC: order-independent fixed-point accumulation, bit-reproducible image).
inline uint32_t flags_from_env() {
    uint32_t f = 0;
    if (const char *e = std::getenv("RTCUDA_WATERTIGHT")) f |= std::atoi(e) != 0 ? RT_FLAG_WATERTIGHT : 0u;
    if (const char *e = std::getenv("RTCUDA_REFERENCE_WALK")) f |= std::atoi(e) != 0 ? RT_FLAG_REFERENCE_WALK : 0u;
    if (const char *e = std::getenv("RTCUDA_DETERMINISTIC")) f |= std::atoi(e) != 0 ? RT_FLAG_DETERMINISTIC : 0u;
    return f;
}

// render.cuh:366-367.  `seed` is the reference's hard-coded RAND_SEED = 1 (render.cuh:417).
inline void render(int width, int height, int num_samples, int max_bounces, Camera camera, Scene scene,
                   std::vector<Vec3> &framebuffer, uint64_t seed = 1, rt_stats *stats = nullptr) {
    rt_scene *h = rtcuda_detail::realise(scene);
    framebuffer.resize((size_t)width * height);
    static_assert(sizeof(Vec3) == 12, "Vec3 must be three packed floats");
    const std::vector<int> devices = devices_from_env();
    const uint32_t flags = flags_from_env();
    if (!devices.empty()) {
        rtcuda_detail::check(rt_render_multi(h, &camera.pod, width, height, num_samples, max_bounces, seed, flags, devices.data(),
                                             (int)devices.size(), reinterpret_cast<float *>(framebuffer.data()), stats), "render");
        return;
    }
    rtcuda_detail::check(rt_render(h, &camera.pod, width, height, num_samples, max_bounces, seed, flags,
                                   reinterpret_cast<float *>(framebuffer.data()), stats), "render");
}

// The same call over several GPUs of the node (rt_render_multi): slot-range shards, one host thread per device, the
// shards' sums added on devices[0].  No reference counterpart -- render.cuh:366-367 is one device.  `devices` are HIP device
// ordinals (their number must divide 1048576); `flags`: RT_FLAG_* of rtcuda_amd.h.
inline void render(int width, int height, int num_samples, int max_bounces, Camera camera, Scene scene,
                   std::vector<Vec3> &framebuffer, const std::vector<int> &devices, uint64_t seed = 1, rt_stats *stats = nullptr,
                   uint32_t flags = 0) {
    rt_scene *h = rtcuda_detail::realise(scene);
    framebuffer.resize((size_t)width * height);
    rtcuda_detail::check(rt_render_multi(h, &camera.pod, width, height, num_samples, max_bounces, seed, flags, devices.data(),
                                         (int)devices.size(), reinterpret_cast<float *>(framebuffer.data()), stats), "render");
}

#endif  // RTCUDA_HPP
