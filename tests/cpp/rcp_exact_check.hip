// rcp_exact_check.hip -- is rt::rcp_exact_normal (rt_device.h) the correctly rounded 1 / x on THIS chip?
//
// The default kernels check every closest hit against the reference's slab test (ref_visible, rtcuda_amd.hip), which
// needs 1 / d exactly as the reference computes it (an IEEE division, aabb_intersector.cuh:17-19).  The product uses
// v_rcp_f32 + one FMA Newton step there (rcp_exact_normal<1>); v_rcp_f32 is a hardware approximation, so the only proof of "same bits" is to
// try every operand the call site can see: the reference clamps |d| to >= FLT_EPSILON and d is a unit vector's component,
// so FLT_EPSILON <= |x| <= 1 -- scanned far wider: EVERY normal fp32 with |x| < 2^126 (biased exponents 1 .. 252, 2 x 2.1 G bit
// patterns), so that the stage-level entry points may be given any finite direction a caller could reasonably pass.  (Beyond
// that range both forms fail: for |x| >= 2^126 the quotient is denormal, and zero / denormal / inf / NaN operands need the
// special-case handling of the compiler's expansion.)
// Built with the product's flags (-ffp-contract=off -fno-fast-math) by tests/test_gpu_parity.py; prints
//   steps1_mismatches steps2_mismatches patterns first_bad_bits_steps2
// and exits 0 iff neither form ever differs from the compiler's IEEE division.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../rtcuda_amd/csrc/rt_device.h"

__global__ void k_scan(uint32_t lo, uint32_t hi, unsigned long long *out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long bad1 = 0, bad2 = 0, first = ~0ull;
    for (uint32_t b = lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi && b >= lo; b += stride) {  // (b >= lo: wrap guard)
        for (int sign = 0; sign < 2; sign++) {
            const uint32_t bits = b | (sign ? 0x80000000u : 0u);
            const float x = __uint_as_float(bits);
            const float want = 1.f / x;
            if (__float_as_uint(rt::rcp_exact_normal<1>(x)) != __float_as_uint(want)) bad1++;
            if (__float_as_uint(rt::rcp_exact_normal<2>(x)) != __float_as_uint(want)) {
                bad2++;
                if (first == ~0ull) first = bits;
            }
        }
    }
    if (bad1) atomicAdd(&out[0], bad1);
    if (bad2) atomicAdd(&out[1], bad2);
    if (first != ~0ull) atomicMin(&out[2], first);
}

int main() {
    const uint32_t lo = 0x00800000u /* 2^-126, the smallest normal */, hi = 0x7e7fffffu /* just below 2^126 */;
    unsigned long long *d = nullptr, h[3] = {0, 0, ~0ull};
    if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) return 2;
    if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_scan, dim3(4096), dim3(256), 0, nullptr, lo, hi, d);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("%llu %llu %llu %llx\n", h[0], h[1], 2ull * (unsigned long long)(hi - lo + 1), h[2]);
    return (h[0] == 0 && h[1] == 0) ? 0 : 1;
}
