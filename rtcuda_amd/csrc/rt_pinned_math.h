// rt_pinned_math.h -- transcendental functions with PINNED fp32 arithmetic.
//
// The reference calls libm/libdevice `sincosf` (utility.cuh:75) and `powf(x, 5)`
// (material.cuh:92). Those are implementation-defined to within a few ulp and differ
// between glibc, CUDA libdevice and ROCm's ocml, so a bit-for-bit CPU<->GPU parity
// check is impossible through them. This header DEFINES the two functions as fixed
// sequences of individually rounded fp32 operations (no FMA: every translation unit that
// includes it is compiled with -ffp-contract=off), so the HIP kernels and the CPU oracle
// produce identical bits for identical inputs. Accuracy is ~1 ulp on the ranges used
// (phi in [0, 2*pi], x in [0, 1]) -- the same class as the libraries they stand in for.
//
// Plain C subset; usable from g++ (host/oracle) and hipcc (device).
#ifndef RT_PINNED_MATH_H
#define RT_PINNED_MATH_H

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD static inline
#endif

// sin/cos of x for x in [-8, 8] (used with x = 2*pi*u, u in (0, 1]).
// Cody-Waite reduction by pi/2 in three fp32 pieces, then the classic single-precision
// minimax polynomials on [-pi/4, pi/4]. Every operation is a separately rounded fp32 op.
RT_HD void rt_sincosf(float x, float *s, float *c) {
    float kf = floorf(x * 0.636619772f + 0.5f);  // nearest multiple of pi/2
    int k = (int)kf;
    float r = x - kf * 1.5703125f;               // exact: 8-bit constant x small integer
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float r2 = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * r2 + 8.3321608736e-3f;
    ps = ps * r2 + -1.6666654611e-1f;
    float sr = r + r * (r2 * ps);
    float pc = 2.443315711809948e-5f;
    pc = pc * r2 + -1.388731625493765e-3f;
    pc = pc * r2 + 4.166664568298827e-2f;
    float cr = (1.0f - 0.5f * r2) + (r2 * r2) * pc;
    float ss = (k & 1) ? cr : sr;
    float cc = (k & 1) ? sr : cr;
    if (k & 2) ss = -ss;
    if ((k + 1) & 2) cc = -cc;
    *s = ss;
    *c = cc;
}

// x^5 as three multiplications in a fixed order (stands in for powf(x, 5)).
RT_HD float rt_pow5f(float x) {
    float x2 = x * x;
    float x4 = x2 * x2;
    return x4 * x;
}

#endif  // RT_PINNED_MATH_H
