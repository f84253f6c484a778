// rt_device.h -- device-side math, sampling and BSDF code of the render path (gfx950).
//
// Every function that feeds the image is written as an explicit sequence of fp32 operations in
// the evaluation order the reference's expressions have (vec3.cuh operator forms, SURVEY.md
// Appendix A.8); the translation unit is compiled with -ffp-contract=off, so these are the same
// roundings the CPU oracle performs and per-ray results can be compared bit for bit.  (The box
// tests in rtcuda_amd.hip only cull and are free to use any conservative arithmetic.)
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_pinned_math.h"

#define RT_DEV __device__ __forceinline__
// tools/isa_census.py builds with -DRT_ISA_MARKS: assembler comments that say which source region an instruction belongs to
// (profiles/r05_adv_census.md).  Nothing in a product build.
#ifdef RT_ISA_MARKS
#define RT_MARK(name) __asm__ volatile("; RT_MARK " name)
#else
#define RT_MARK(name)
#endif

namespace rt {

constexpr float kInvPi = 0.31830988618379067153f;  // constant.hpp:6
constexpr float kTwoPi = 6.28318530717958647692f;  // constant.hpp:5
constexpr int kRrStart = 4;                        // constant.hpp:10
constexpr float kRrThreshold = 1.f;                // constant.hpp:9
constexpr float kFltMax = 3.402823466e+38f;
constexpr float kFltEps = 1.192092896e-07f;

struct V3 {
    float x, y, z;
};
RT_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
RT_DEV V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 scale(V3 a, float t) { return mk(a.x * t, a.y * t, a.z * t); }
RT_DEV V3 divf(V3 a, float t) {  // vec3.cuh:56-59: multiply by the reciprocal
    float inv_t = 1.f / t;
    return mk(a.x * inv_t, a.y * inv_t, a.z * inv_t);
}
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_DEV float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_DEV float len(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
RT_DEV V3 unit(V3 a) {
    float inv_len = 1.f / len(a);
    return mk(a.x * inv_len, a.y * inv_len, a.z * inv_len);
}
RT_DEV float max3(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
// 1 / x, CORRECTLY ROUNDED for every NORMAL x with |x| < 2^126: v_rcp_f32 (1 ulp) and STEPS Newton steps in FMA arithmetic
// -- 1 + 2 * STEPS instructions instead of the 11 of the compiler's IEEE division, which must also serve zero, denormals,
// infinities, NaN and quotients that underflow.  tests/cpp/rcp_exact_check.hip compares the one-step and the two-step form
// with `1.f / x` for EVERY fp32 bit pattern of that range on the GPU (4.2 * 10^9 operands; v_rcp_f32 is this chip's, not a
// formula): none differs, for either form; the GPU suite repeats the scan.  Only for call sites that can PROVE the range: the
// slab setup of ref_visible, whose operand is a unit vector's component with FLT_EPSILON <= |x|.  (The estimator's own
// reciprocals -- 1 / dot(d, n), 1 / pdf, 1 / length -- can be zero, denormal or arbitrarily small: with a class test and a
// rare branch to the compiler's form the saving is 2 - 3 instructions per site, ~0.5 % of the frame: not built.)
template <int STEPS = 1>
RT_DEV float rcp_exact_normal(float x) {
    float r = __builtin_amdgcn_rcpf(x);
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
        const float e = fmaf(-x, r, 1.f);
        r = fmaf(e, r, r);
    }
    return r;
}
RT_DEV V3 reflect(V3 v, V3 n) { return sub(v, scale(n, 2.f * dot(v, n))); }  // vec3.cuh:71-73
RT_DEV V3 refract4(V3 unit_v, V3 unit_n, float eta_ratio, float cos_theta) {  // vec3.cuh:82-86
    V3 v_parallel = scale(add(unit_v, scale(unit_n, cos_theta)), eta_ratio);
    V3 v_perp = scale(unit_n, -sqrtf(1.f - len2(v_parallel)));
    return add(v_parallel, v_perp);
}

// ------------------------------------------------------------------ XORWOW (cuRAND's generator)
struct Rng {
    uint32_t d, v0, v1, v2, v3, v4;
};
RT_DEV uint32_t rng_next(Rng &s) {
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1;
    s.v1 = s.v2;
    s.v2 = s.v3;
    s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}
// RT_FLAG_RNG_PER_SAMPLE: the stream of ONE camera ray, keyed by (seed, global camera-ray id) -- not the reference's
// stream (there a path continues the stream of its SLOT: render.cuh:72,156,263), so the image is a different, statistically
// equivalent estimate; in exchange it does not depend on which slot, or which GPU, serves the camera ray.  splitmix64 of
// the key, then curand_init's seed scramble of that word as the XORWOW state, subsequence 0.
RT_DEV Rng rng_sample_stream(uint32_t seed_lo, uint32_t seed_hi, unsigned long long key) {
    unsigned long long z = (((unsigned long long)seed_hi << 32) | seed_lo) + 0x9E3779B97F4A7C15ull * (key + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const uint32_t s0 = (uint32_t)z ^ 0xaad26b49u, s1 = (uint32_t)(z >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    Rng r;
    r.d = 6615241u + t1 + t0;
    r.v0 = 123456789u + t0;
    r.v1 = 362436069u ^ t0;
    r.v2 = 521288629u + t1;
    r.v3 = 88675123u ^ t1;
    r.v4 = 5783321u + t0;
    return r;
}
// curand_uniform: (0, 1]
RT_DEV float rng_uniform(Rng &s) { return (float)rng_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }

// ------------------------------------------------------------------ triangle (triangle.cuh)
struct Tri {
    V3 p0, e1, e2, n;
};
// 48-byte record = 3 x float4: (p0.xyz, e1.x) (e1.yz, e2.xy) (e2.z, n.xyz)
RT_DEV Tri load_tri(const float4 *__restrict__ tris, int k) {
    // 32-bit byte offset from a uniform base: lets the compiler use the SGPR-base addressing form
    // (24-bit multiply: full rate, the 32-bit one is quarter rate; a scene has far fewer than 2^24 triangles -- checked
    // by rt_scene_create)
    const float4 *q = (const float4 *)((const char *)tris + __umul24((unsigned)k, 48u));
    float4 a = q[0];
    float4 b = q[1];
    float4 c = q[2];
    Tri t;
    t.p0 = mk(a.x, a.y, a.z);
    t.e1 = mk(a.w, b.x, b.y);
    t.e2 = mk(b.z, b.w, c.x);
    t.n = mk(c.y, c.z, c.w);
    return t;
}
// triangle.cuh:39-58 -- the ONE triangle test; closest-hit, any-hit and Light::pdf_Li all use it
RT_DEV bool tri_intersect(const Tri &tr, V3 o, V3 d, float tmax, float &t_out, float &u_out, float &v_out) {
    V3 c = sub(tr.p0, o);
    V3 r = cross(d, c);
    float inv_det = 1.f / dot(d, tr.n);
    float u = inv_det * dot(tr.e2, r);
    float v = inv_det * dot(tr.e1, r);
    // (t for every lane, and one straight-line result: in a wave some lane nearly always passes the barycentric test, so
    // the nested early exits of the reference's form save nothing here and cost two exec-mask regions and their merges;
    // the values and the accept / reject decision are the same.  t, u, v are only meaningful when true is returned.)
    float t = inv_det * dot(c, tr.n);
    t_out = t;
    u_out = u;
    v_out = v;
    return u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f && 0 < t && t <= tmax;
}
RT_DEV V3 tri_point(const Tri &tr, float u, float v) { return add(sub(tr.p0, scale(tr.e1, u)), scale(tr.e2, v)); }
RT_DEV float tri_area(const Tri &tr) { return 0.5f * len(tr.n); }  // 0.5 * x is exact in any precision

// ------------------------------------------------------------------ utility.cuh
RT_DEV V3 offset_ray_origin(V3 p, V3 n) {  // :31-47
    const float int_scale = 256.f;
    const float float_scale = 1.f / 65536.f;
    const float origin = 1.f / 32.f;
    int ox = (int)(int_scale * n.x);
    int oy = (int)(int_scale * n.y);
    int oz = (int)(int_scale * n.z);
    float px = __int_as_float(__float_as_int(p.x) + (p.x < 0 ? -ox : ox));
    float py = __int_as_float(__float_as_int(p.y) + (p.y < 0 ? -oy : oy));
    float pz = __int_as_float(__float_as_int(p.z) + (p.z < 0 ? -oz : oz));
    return mk(fabsf(p.x) < origin ? p.x + float_scale * n.x : px,
              fabsf(p.y) < origin ? p.y + float_scale * n.y : py,
              fabsf(p.z) < origin ? p.z + float_scale * n.z : pz);
}
// :53-56 with the int parameter made explicit; only called where g < 2^31 (NEE: g = cos/pi < 1)
RT_DEV float power_heuristic(float f_pdf, float g_pdf_float) {
    int g = (int)g_pdf_float;
    float f2 = f_pdf * f_pdf;
    return f2 / (f2 + (float)(g * g));
}
RT_DEV bool same_hemisphere(V3 wo, V3 wi, V3 n) { return dot(wo, n) * dot(wi, n) < 0.f; }  // :58-60
RT_DEV V3 uniform_sample_sphere(Rng &rs) {  // :70-77
    float z = 1 - 2 * rng_uniform(rs);
    float r = sqrtf(1 - z * z);
    float phi = kTwoPi * rng_uniform(rs);
    float x, y;
    rt_sincosf(phi, &y, &x);
    return mk(r * x, r * y, z);
}

// ------------------------------------------------------------------ material.cuh
struct Material {
    float ax, ay, az;  // albedo
    float ior;
    int type;  // 0 MATTE 1 MIRROR 2 GLASS
};
RT_DEV bool mat_get_f(const Material &m, V3 wo, V3 wi, V3 n, V3 &f, float &pdf) {  // :47-57
    if (m.type == 0) {
        if (same_hemisphere(wo, wi, n)) {
            f = scale(mk(m.ax, m.ay, m.az), kInvPi);
            pdf = dot(wi, n) * kInvPi;
            return true;
        }
    }
    return false;
}
// :60-109 -- mutates n so that n and wi share a hemisphere.
// `again_draws`: how many uniforms a SECOND call with the same (material, wo, n) consumes -- mat()'s "sample BSDF with MIS"
// block makes one (render.cuh:218: same unit_wo, isect_unit_n again) whose ray can never be credited (Appendix A.3), so
// only its effect on the RNG stream is kept, and that is decided by values this call computes anyway: 2 for matte
// (uniform_sample_sphere), 0 for a mirror, and for glass 1 unless the ray cannot refract (the Schlick draw :94).  (Until
// round 4 a second function recomputed cos / sin / the refraction test for that: 60 vector instructions per shade.  The
// oracle makes the second call in full, as the reference does: every parity test holds the two against each other.)
RT_DEV V3 mat_sample_f(const Material &m, V3 wo, Rng &rs, V3 &n, V3 &wi, float &pdf, int &again_draws) {
    again_draws = 0;
    if (m.type == 0 || m.type == 1) {
        if (dot(wo, n) > 0.f) n = neg(n);
        if (m.type == 0) {
            again_draws = 2;
            wi = unit(add(n, uniform_sample_sphere(rs)));
            pdf = dot(wi, n) * kInvPi;
            return scale(mk(m.ax, m.ay, m.az), kInvPi);
        } else {
            wi = reflect(wo, n);
            pdf = 1.f;
            return divf(mk(m.ax, m.ay, m.az), dot(wi, n));
        }
    } else {
        float cos_theta = dot(wo, n);
        bool front = cos_theta < 0.f;
        if (front) cos_theta = -cos_theta;
        float inv_cos = 1.f / cos_theta;
        float eta = front ? 1.f / m.ior : m.ior;
        float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
        bool cannot_refract = eta * sin_theta > 1.f;
        if (cannot_refract) {
            if (!front) n = neg(n);
            wi = reflect(wo, n);
            pdf = 1.f;
            return mk(inv_cos, inv_cos, inv_cos);
        }
        again_draws = 1;
        float r0 = (1 - m.ior) / (1 + m.ior);
        r0 = r0 * r0;
        float reflectance = r0 + (1 - r0) * rt_pow5f(1 - cos_theta);
        if (rng_uniform(rs) < reflectance) {
            if (!front) n = neg(n);
            wi = reflect(wo, n);
            pdf = reflectance;
            float v = pdf * inv_cos;
            return mk(v, v, v);
        } else {
            if (!front) n = neg(n);
            wi = refract4(wo, n, eta, cos_theta);
            n = neg(n);
            pdf = 1.f - reflectance;
            float v = pdf * eta * eta / dot(wi, n);
            return mk(v, v, v);
        }
    }
}
// ------------------------------------------------------------------ light.cuh
struct Light {
    int type;  // 0 POINT 1 AREA
    float px, py, pz;
    int tri;  // leaf-order triangle index
    float lx, ly, lz;
};

// ------------------------------------------------------------------ camera.cuh:31-34
struct Camera {
    V3 lookfrom, upper_left, horizontal, vertical;
};
RT_DEV void camera_get_ray(const Camera &c, float x, float y, V3 &o, V3 &d) {
    V3 dir = sub(add(add(c.upper_left, scale(c.horizontal, x)), scale(c.vertical, y)), c.lookfrom);
    o = c.lookfrom;
    d = unit(dir);
}

}  // namespace rt
#endif  // RT_DEVICE_H
