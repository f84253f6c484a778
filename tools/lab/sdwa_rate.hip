// Lab: issue rate of the candidate 16-bit -> fp32 decode instructions on gfx950 (4 waves per SIMD, 1024 x 256 threads).
//   hipcc --offload-arch=gfx950 -O2 -o tools/lab/sdwa_rate tools/lab/sdwa_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned seed) {
    unsigned w0 = seed + threadIdx.x, w1 = w0 * 3u, w2 = w0 * 5u, w3 = w0 * 7u;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            asm volatile("v_cvt_f32_u32_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                         "v_cvt_f32_u32_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                         "v_cvt_f32_u32_sdwa %2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                         "v_cvt_f32_u32_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(w0), "v"(w1));
            asm volatile("v_cvt_f32_u32_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                         "v_cvt_f32_u32_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                         "v_cvt_f32_u32_sdwa %2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                         "v_cvt_f32_u32_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                         : "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(w2), "v"(w3));
        } else if (KIND == 1) {
            asm volatile("v_cvt_f32_u32_e32 %0, %4\nv_cvt_f32_u32_e32 %1, %5\nv_cvt_f32_u32_e32 %2, %4\nv_cvt_f32_u32_e32 %3, %5\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(w0), "v"(w1));
            asm volatile("v_cvt_f32_u32_e32 %0, %4\nv_cvt_f32_u32_e32 %1, %5\nv_cvt_f32_u32_e32 %2, %4\nv_cvt_f32_u32_e32 %3, %5\n"
                         : "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(w2), "v"(w3));
        } else if (KIND == 2) {
            asm volatile("v_and_or_b32 %0, %4, %6, %7\nv_and_or_b32 %1, %5, %6, %7\nv_and_or_b32 %2, %4, %6, %7\nv_and_or_b32 %3, %5, %6, %7\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(w0), "v"(w1), "s"(0xffffu), "v"(0x4b000000u));
            asm volatile("v_and_or_b32 %0, %4, %6, %7\nv_and_or_b32 %1, %5, %6, %7\nv_and_or_b32 %2, %4, %6, %7\nv_and_or_b32 %3, %5, %6, %7\n"
                         : "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(w2), "v"(w3), "s"(0xffffu), "v"(0x4b000000u));
        } else if (KIND == 3) {
            asm volatile("v_perm_b32 %0, %4, %6, %7\nv_perm_b32 %1, %5, %6, %7\nv_perm_b32 %2, %4, %6, %7\nv_perm_b32 %3, %5, %6, %7\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(w0), "v"(w1), "v"(0x4b000000u), "s"(0x03020706u));
            asm volatile("v_perm_b32 %0, %4, %6, %7\nv_perm_b32 %1, %5, %6, %7\nv_perm_b32 %2, %4, %6, %7\nv_perm_b32 %3, %5, %6, %7\n"
                         : "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(w2), "v"(w3), "v"(0x4b000000u), "s"(0x03020706u));
        } else if (KIND == 4) {
            asm volatile("v_mul_f32_e32 %0, %4, %4\nv_mul_f32_e32 %1, %5, %5\nv_mul_f32_e32 %2, %4, %5\nv_mul_f32_e32 %3, %5, %4\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(w0), "v"(w1));
            asm volatile("v_mul_f32_e32 %0, %4, %4\nv_mul_f32_e32 %1, %5, %5\nv_mul_f32_e32 %2, %4, %5\nv_mul_f32_e32 %3, %5, %4\n"
                         : "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(w2), "v"(w3));
        } else if (KIND == 5) {
            asm volatile("v_cvt_f32_ubyte0_e32 %0, %4\nv_cvt_f32_ubyte1_e32 %1, %4\nv_cvt_f32_ubyte2_e32 %2, %5\nv_cvt_f32_ubyte3_e32 %3, %5\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(w0), "v"(w1));
            asm volatile("v_cvt_f32_ubyte0_e32 %0, %4\nv_cvt_f32_ubyte1_e32 %1, %4\nv_cvt_f32_ubyte2_e32 %2, %5\nv_cvt_f32_ubyte3_e32 %3, %5\n"
                         : "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(w2), "v"(w3));
        }
        w0 += __float_as_uint(a0) & 1u; w1 ^= i; w2 += 3; w3 ^= __float_as_uint(a7) & 2u;
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, float *out) {
    const int iters = 20000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(256), 0, 0, out, iters, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(256), 0, 0, out, iters, 2u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // 8 probe instructions + ~5 bookkeeping per iteration; per SIMD: 4 waves
    const double instr_per_simd = 4.0 * iters * 8;
    printf("%-28s %8.3f ms   %6.2f cycles per probe instruction per SIMD (2.4 GHz, bookkeeping included)\n", name, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main() {
    float *out; CK(hipMalloc(&out, 1024 * 256 * 4));
    run<4>("v_mul_f32", out);
    run<1>("v_cvt_f32_u32", out);
    run<0>("v_cvt_f32_u32_sdwa WORD", out);
    run<2>("v_and_or_b32", out);
    run<3>("v_perm_b32", out);
    run<5>("v_cvt_f32_ubyteN", out);
    return 0;
}
