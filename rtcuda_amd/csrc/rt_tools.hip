// rt_tools.hip -- the LAB: measurement tools that are not part of the product's C-ABI (include/rtcuda_amd_tools.h).
//
// librtcuda_amd_tools.so is built from this file, which includes the product's translation unit -- so the tools measure the
// product's own kernels (the split probe runs k_advance / k_trace and the product's advance_core on dumped rays) -- and adds:
//   rt_measure_copy_bandwidth   device copy bandwidth (the HBM roof SURVEY 8d wants measured in the same run)
//   rt_calibrate_valu(_packed)  what the vector ALUs sustain on independent v_fma_f32 / v_pk_* streams
//   rt_probe_issue              cycles per instruction of one wave, by instruction kind and occupancy
//   rt_split_probe              "one persistent kernel, or the reference's stage split?" priced on dense ray / shade arrays
// The product library (librtcuda_amd.so) contains none of this.  The tools library also carries a private copy of the
// product's entry points (same source); a scene handed to rt_split_probe must come from THIS library's rt_scene_create
// (rtcuda_amd/api.py: Scene(arrays, library=tools_lib())).
#include "rtcuda_amd.hip"
#include "../../include/rtcuda_amd_tools.h"

__global__ void k_copy_f4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

// VALU issue calibration: every wave issues `iters` x 16 independent v_fma_f32 (inline asm, so the compiler neither
// packs nor folds them).  Launched with `waves_per_simd` waves on every SIMD it measures what the vector ALU of this
// chip sustains in lane-operations per second -- the roof the render kernels (VALU-issue-bound) are priced against --
// and, run under the PMC set of tools/, what SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES read at that known rate.
__global__ void __launch_bounds__(256) k_valu_calibrate(float *__restrict__ out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1.f, a2 = seed + 2.f, a3 = seed + 3.f, a4 = seed + 4.f, a5 = seed + 5.f, a6 = seed + 6.f,
          a7 = seed + 7.f, a8 = seed + 8.f, a9 = seed + 9.f, a10 = seed + 10.f, a11 = seed + 11.f, a12 = seed + 12.f,
          a13 = seed + 13.f, a14 = seed + 14.f, a15 = seed + 15.f;
    const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
    for (int k = 0; k < iters; k++) {
        __asm__ volatile(
            "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
            "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
            "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
            "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10),
              "+v"(a11), "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15)
            : "v"(m), "v"(c));
    }
    float r = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + ((a8 + a9) + (a10 + a11)) + ((a12 + a13) + (a14 + a15));
    if (r == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r;  // (never true: keeps the chain alive)
}

// The same stream in packed fp32 (KIND 1: v_pk_fma_f32, 2: v_pk_mul_f32, 3: v_pk_add_f32): 16 independent chains on 16
// aligned register PAIRS, two lane-operations per lane and instruction.  Answers one question before any hand-packing of
// the shading arithmetic: does a packed instruction issue at the rate of a scalar one (2x the lane-operations), or at half?
template <int KIND>
__global__ void __launch_bounds__(256) k_valu_calibrate_pk(float *__restrict__ out, int iters, float seed) {
    v2f a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = v2f{seed + (float)k, seed + 0.5f + (float)k};
    const v2f m = {0.999f + seed * 1e-9f, 0.998f + seed * 1e-9f}, c = {1e-3f, 2e-3f};
    for (int it = 0; it < iters; it++) {
#define RT_PK16(OP, ARGS)                                                                                              \
    __asm__ volatile(OP " %0, %0, " ARGS "\n" OP " %1, %1, " ARGS "\n" OP " %2, %2, " ARGS "\n" OP " %3, %3, " ARGS "\n"     \
                     OP " %4, %4, " ARGS "\n" OP " %5, %5, " ARGS "\n" OP " %6, %6, " ARGS "\n" OP " %7, %7, " ARGS "\n"     \
                     OP " %8, %8, " ARGS "\n" OP " %9, %9, " ARGS "\n" OP " %10, %10, " ARGS "\n" OP " %11, %11, " ARGS "\n" \
                     OP " %12, %12, " ARGS "\n" OP " %13, %13, " ARGS "\n" OP " %14, %14, " ARGS "\n" OP " %15, %15, " ARGS  \
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),      \
                       "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                     : "v"(m), "v"(c))
        if (KIND == 1) RT_PK16("v_pk_fma_f32", "%16, %17");
        else if (KIND == 2) RT_PK16("v_pk_mul_f32", "%16");
        else RT_PK16("v_pk_add_f32", "%17");
#undef RT_PK16
    }
    v2f r = a[0];
#pragma unroll
    for (int k = 1; k < 16; k++) r = r + a[k];
    if (r.x + r.y == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r.x;  // (never true: keeps the chains alive)
}

// Issue probe (rt_probe_issue): the calibration stream with OTHER instructions -- how long one wave needs per instruction
// of a given kind, alone on its SIMD and beside 1 / 3 / 7 other waves.  A block of k_paths takes the same time whether its
// SIMD holds one wave or four, i.e. the kernel is bound by what ONE wave can issue, and that depends on
// the instruction: this probe is how it was measured.  16 independent chains per lane unless the kind says "chain".
#define RT_P16(L) L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15)
#define RT_PL_FMA(k) "v_fma_f32 %" #k ", %" #k ", %16, %17\n"
#define RT_PL_FMAC(k) "v_fmac_f32 %" #k ", %16, %17\n"
#define RT_PL_MUL(k) "v_mul_f32 %" #k ", %16, %" #k "\n"
#define RT_PL_ADD(k) "v_add_f32 %" #k ", %17, %" #k "\n"
#define RT_PL_MOV(k) "v_mov_b32 %" #k ", %16\n"
#define RT_PL_XOR(k) "v_xor_b32 %" #k ", %16, %" #k "\n"
#define RT_PL_SHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define RT_PL_MAX(k) "v_max_f32 %" #k ", %16, %" #k "\n"
#define RT_PL_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define RT_PL_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n"
#define RT_PL_CND(k) "v_cndmask_b32 %" #k ", %" #k ", %16, vcc\n"
#define RT_PL_MULADD(k) "v_mul_f32 %" #k ", %16, %" #k "\n v_add_f32 %" #k ", %17, %" #k "\n"
#define RT_PL_CHAIN_FMA(k) "v_fma_f32 %0, %0, %16, %17\n"
#define RT_PL_CHAIN_MUL(k) "v_mul_f32 %0, %16, %0\n"
#define RT_PL_MUL_LIT(k) "v_mul_f32 %" #k ", 0x3f7fbe77, %" #k "\n"
#define RT_PL_MUL_SGPR(k) "v_mul_f32 %" #k ", %18, %" #k "\n"
#define RT_PL_FMA_SGPR(k) "v_fma_f32 %" #k ", %" #k ", %18, %19\n"
#define RT_PL_CND_SGPR(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %16, %20\n"
#define RT_PL_CMP(k) "v_cmp_lt_f32_e32 vcc, %16, %" #k "\n"
#define RT_PL_CMP_CND(k) "v_cmp_lt_f32_e32 vcc, %17, %" #k "\n v_cndmask_b32_e32 %" #k ", %" #k ", %16, vcc\n"
#define RT_PL_BFI(k) "v_bfi_b32 %" #k ", %16, %17, %" #k "\n"
#define RT_PL_AND(k) "v_and_b32_e32 %" #k ", %16, %" #k "\n"
#define RT_PL_MIX_CND(k) "v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %16, %" #k "\n v_cndmask_b32_e32 %" #k ", %" #k ", %17, vcc\n"
#define RT_PL_MIX_MAX(k) "v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %16, %" #k "\n v_max_f32_e32 %" #k ", %17, %" #k "\n"
#define RT_PL_MIX_MUL(k) "v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %17, %" #k "\n"
#define RT_PL_MIX_MOV(k) "v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mul_f32_e32 %" #k ", %16, %" #k "\n v_mov_b32_e32 %" #k ", %" #k "\n"
enum { kProbeKinds = 26 };
template <int KIND>
__global__ void __launch_bounds__(256) k_probe_issue(float *__restrict__ out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1.f, a2 = seed + 2.f, a3 = seed + 3.f, a4 = seed + 4.f, a5 = seed + 5.f, a6 = seed + 6.f,
          a7 = seed + 7.f, a8 = seed + 8.f, a9 = seed + 9.f, a10 = seed + 10.f, a11 = seed + 11.f, a12 = seed + 12.f,
          a13 = seed + 13.f, a14 = seed + 14.f, a15 = seed + 15.f;
    const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
    const float sm = __builtin_amdgcn_readfirstlane(m), sc = __builtin_amdgcn_readfirstlane(c);
    const unsigned long long lane_mask = __builtin_amdgcn_ballot_w64(seed + (float)(threadIdx.x & 1) > 1.5f);  // (an SGPR pair)
#define RT_PROBE_ASM(L)                                                                                                   \
    __asm__ volatile(RT_P16(L)                                                                                            \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9),     \
                       "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15)                                       \
                     : "v"(m), "v"(c), "s"(sm), "s"(sc), "s"(lane_mask)                                                          \
                     : "vcc")
    for (int k = 0; k < iters; k++) {
        if (KIND == 0) RT_PROBE_ASM(RT_PL_FMA);
        else if (KIND == 1) RT_PROBE_ASM(RT_PL_FMAC);
        else if (KIND == 2) RT_PROBE_ASM(RT_PL_MUL);
        else if (KIND == 3) RT_PROBE_ASM(RT_PL_ADD);
        else if (KIND == 4) RT_PROBE_ASM(RT_PL_MOV);
        else if (KIND == 5) RT_PROBE_ASM(RT_PL_XOR);
        else if (KIND == 6) RT_PROBE_ASM(RT_PL_SHL);
        else if (KIND == 7) RT_PROBE_ASM(RT_PL_MAX);
        else if (KIND == 8) RT_PROBE_ASM(RT_PL_RCP);
        else if (KIND == 9) RT_PROBE_ASM(RT_PL_SQRT);
        else if (KIND == 10) RT_PROBE_ASM(RT_PL_CND);
        else if (KIND == 11) RT_PROBE_ASM(RT_PL_MULADD);
        else if (KIND == 12) RT_PROBE_ASM(RT_PL_CHAIN_FMA);
        else if (KIND == 13) RT_PROBE_ASM(RT_PL_CHAIN_MUL);
        else if (KIND == 14) RT_PROBE_ASM(RT_PL_MUL_LIT);
        else if (KIND == 15) RT_PROBE_ASM(RT_PL_MUL_SGPR);
        else if (KIND == 16) RT_PROBE_ASM(RT_PL_FMA_SGPR);
        else if (KIND == 17) RT_PROBE_ASM(RT_PL_CND_SGPR);
        else if (KIND == 18) RT_PROBE_ASM(RT_PL_CMP);
        else if (KIND == 19) RT_PROBE_ASM(RT_PL_CMP_CND);
        else if (KIND == 20) RT_PROBE_ASM(RT_PL_BFI);
        else if (KIND == 21) RT_PROBE_ASM(RT_PL_AND);
        else if (KIND == 22) RT_PROBE_ASM(RT_PL_MIX_CND);
        else if (KIND == 23) RT_PROBE_ASM(RT_PL_MIX_MAX);
        else if (KIND == 24) RT_PROBE_ASM(RT_PL_MIX_MUL);
        else RT_PROBE_ASM(RT_PL_MIX_MOV);
    }
#undef RT_PROBE_ASM
    float r = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + ((a8 + a9) + (a10 + a11)) + ((a12 + a13) + (a14 + a15));
    if (r == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r;  // (never true: keeps the chains alive)
}

// ============================================================================ split probe (rt_split_probe)
// Measurement kernels for the question "would separate trace and shade kernels -- the reference's stage split
// (render.cuh:428-449) with dense queues -- beat k_paths?".  The round pipeline's pools are copied, round after round,
// into DENSE arrays: the rays k_trace is about to trace (closest-hit and any-hit apart), and the slot records
// k_advance is about to shade (one bucket per material kind: a wave of the shading probe sees one material).
// The trace side is then timed with k_trace's stage-level modes on those arrays, the shade side with k_probe_shade.
constexpr int kProbeIn = 22;   // dwords of a shading record: bounces, hit_info, pixel, gen, rng 6, beta 3, wo 3, p 3, n 3
constexpr int kProbeOut = 27;  // dwords a shade writes: ray 6, shadow ray 6 + tmax + L 3 + target, beta 3, rng 6, bounces
__global__ void __launch_bounds__(kBlock)
k_probe_dump_rays(DPools p, int n, float *__restrict__ c_o3, float *__restrict__ c_d3, float *__restrict__ c_tmax,
                  float *__restrict__ a_o3, float *__restrict__ a_d3, float *__restrict__ a_tmax, int *__restrict__ a_excl,
                  unsigned cap, unsigned *__restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const int b = in ? p.bounces(i) : kDone;
    const bool live = in && b != kDone && b != kParked;
    const bool shadow = in && p.stmax(i) >= 0.f;
    unsigned long long m = wave_ballot(live);
    if (m) {
        unsigned base = 0;
        if (lane_id() == 0) base = atomicAdd(&counts[0], (unsigned)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned k = base + prefix_popc(m);
        if (live && k < cap) {
            c_o3[3 * (size_t)k + 0] = p.ox(i);
            c_o3[3 * (size_t)k + 1] = p.oy(i);
            c_o3[3 * (size_t)k + 2] = p.oz(i);
            c_d3[3 * (size_t)k + 0] = p.dx(i);
            c_d3[3 * (size_t)k + 1] = p.dy(i);
            c_d3[3 * (size_t)k + 2] = p.dz(i);
            c_tmax[k] = kFltMax;
        }
    }
    m = wave_ballot(shadow);
    if (m) {
        unsigned base = 0;
        if (lane_id() == 0) base = atomicAdd(&counts[1], (unsigned)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned k = base + prefix_popc(m);
        if (shadow && k < cap) {
            a_o3[3 * (size_t)k + 0] = p.sox(i);
            a_o3[3 * (size_t)k + 1] = p.soy(i);
            a_o3[3 * (size_t)k + 2] = p.soz(i);
            a_d3[3 * (size_t)k + 0] = p.sdx(i);
            a_d3[3 * (size_t)k + 1] = p.sdy(i);
            a_d3[3 * (size_t)k + 2] = p.sdz(i);
            a_tmax[k] = p.stmax(i);
            a_excl[k] = p.starget(i);
        }
    }
}
// the slots the NEXT k_advance will shade (hit, bounce left: render.cuh:109,128-130), by material kind
__global__ void __launch_bounds__(kBlock)
k_probe_dump_shades(DScene sc, DPools p, int n, int max_bounces, float *__restrict__ rec, unsigned cap,
                    unsigned *__restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const int b = in ? p.bounces(i) : kDone;
    const int hi = in ? p.hit_info(i) : -1;
    const bool shade = in && b != kDone && b != kParked && hi >= 0 && b < max_bounces;
    const int kind = shade ? __float_as_int(sc.tables[5 * (hi & 0xffff) + 4]) : -1;
    for (int mk = 0; mk < 3; mk++) {
        const bool mine = kind == mk;
        const unsigned long long m = wave_ballot(mine);
        if (!m) continue;
        unsigned base = 0;
        if (lane_id() == 0) base = atomicAdd(&counts[2 + mk], (unsigned)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned k = base + prefix_popc(m);
        if (mine && k < cap) {
            float *r = rec + (size_t)mk * kProbeIn * cap + k;  // array a of bucket mk at r[a * cap]
            const float v[kProbeIn] = {__int_as_float(b), __int_as_float(hi), __int_as_float(p.pixel(i)), __int_as_float(p.gen(i)),
                                       __uint_as_float(p.rd(i)), __uint_as_float(p.r0(i)), __uint_as_float(p.r1(i)),
                                       __uint_as_float(p.r2(i)), __uint_as_float(p.r3(i)), __uint_as_float(p.r4(i)),
                                       p.br(i), p.bg(i), p.bb(i), p.dx(i), p.dy(i), p.dz(i),
                                       p.hpx(i), p.hpy(i), p.hpz(i), p.hnx(i), p.hny(i), p.hnz(i)};
#pragma unroll
            for (int a = 0; a < kProbeIn; a++) r[(size_t)a * cap] = v[a];
        }
    }
}
// init() + mat() (advance_core, the code k_advance and k_paths run) on a dense array of shading records: every lane
// of every wave shades, one material kind per launch.  Reads kProbeIn dwords per shade, writes up to kProbeOut.
template <bool LDS_TABLES>
__global__ void __launch_bounds__(kBlock)
k_probe_shade(DScene sc, Camera cam, AdvanceParams ap, const float *__restrict__ rec, unsigned cap, unsigned count,
              float *__restrict__ outp, float *__restrict__ fb) {
    __shared__ float s_tab[LDS_TABLES ? kTabDwordsMax : 1];
    const float *tab = sc.tables;
    if (LDS_TABLES) {
        for (int k = threadIdx.x; k < sc.tab_dwords; k += kBlock) s_tab[k] = sc.tables[k];
        __syncthreads();
        tab = s_tab;
    }
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float *r = rec + i;
    SlotState st;
    st.bounces = __float_as_int(r[0 * (size_t)cap]);
    st.hit_info = __float_as_int(r[1 * (size_t)cap]);
    st.pixel = __float_as_int(r[2 * (size_t)cap]);
    st.gen = __float_as_int(r[3 * (size_t)cap]);
    st.rs = Rng{__float_as_uint(r[4 * (size_t)cap]), __float_as_uint(r[5 * (size_t)cap]), __float_as_uint(r[6 * (size_t)cap]),
                __float_as_uint(r[7 * (size_t)cap]), __float_as_uint(r[8 * (size_t)cap]), __float_as_uint(r[9 * (size_t)cap])};
    st.beta = mk(r[10 * (size_t)cap], r[11 * (size_t)cap], r[12 * (size_t)cap]);
    st.wo = mk(r[13 * (size_t)cap], r[14 * (size_t)cap], r[15 * (size_t)cap]);
    st.isect_p = mk(r[16 * (size_t)cap], r[17 * (size_t)cap], r[18 * (size_t)cap]);
    st.isect_n = mk(r[19 * (size_t)cap], r[20 * (size_t)cap], r[21 * (size_t)cap]);
    AdvanceOut out;
    advance_core<true, true, false>(sc, tab, cam, ap, 0, st, out, fb);
    float *w = outp + i;
    if (out.new_ray) {
        w[0 * (size_t)cap] = out.ray_o.x;
        w[1 * (size_t)cap] = out.ray_o.y;
        w[2 * (size_t)cap] = out.ray_o.z;
        w[3 * (size_t)cap] = out.ray_d.x;
        w[4 * (size_t)cap] = out.ray_d.y;
        w[5 * (size_t)cap] = out.ray_d.z;
    }
    w[12 * (size_t)cap] = out.has_shadow ? out.s_tmax : -1.f;
    if (out.has_shadow) {
        w[6 * (size_t)cap] = out.s_o.x;
        w[7 * (size_t)cap] = out.s_o.y;
        w[8 * (size_t)cap] = out.s_o.z;
        w[9 * (size_t)cap] = out.s_d.x;
        w[10 * (size_t)cap] = out.s_d.y;
        w[11 * (size_t)cap] = out.s_d.z;
        w[13 * (size_t)cap] = out.s_L.x;
        w[14 * (size_t)cap] = out.s_L.y;
        w[15 * (size_t)cap] = out.s_L.z;
        w[16 * (size_t)cap] = __int_as_float(out.s_target);
    }
    w[17 * (size_t)cap] = st.beta.x;
    w[18 * (size_t)cap] = st.beta.y;
    w[19 * (size_t)cap] = st.beta.z;
    w[20 * (size_t)cap] = __uint_as_float(st.rs.d);
    w[21 * (size_t)cap] = __uint_as_float(st.rs.v0);
    w[22 * (size_t)cap] = __uint_as_float(st.rs.v1);
    w[23 * (size_t)cap] = __uint_as_float(st.rs.v2);
    w[24 * (size_t)cap] = __uint_as_float(st.rs.v3);
    w[25 * (size_t)cap] = __uint_as_float(st.rs.v4);
    w[26 * (size_t)cap] = __int_as_float(st.bounces);
}


// ---- split probe: see the kernels (k_probe_*) for what is measured
namespace {
template <int MODE, bool WIDE, int MINW>
int probe_trace_once(const rt_scene *scene, const TraceParams &tp, int stack_cap, int *d_over, int cus, hipEvent_t e0,
                     hipEvent_t e1, double *seconds, double *blocks_per_cu) {
    const size_t lds = sizeof(int) * kBlock * (size_t)(stack_cap + 2);
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_trace<MODE, WIDE, MINW>, kBlock, lds));
    occ = std::max(1, occ);
    const int grid = std::max(1, std::min(grid_for(tp.total), cus * occ));
    double best = 1e30;
    DPools none{};
    for (int rep = 0; rep < 3; rep++) {
        HIP_TRY(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL((k_trace<MODE, WIDE, MINW>), dim3(grid), dim3(kBlock), lds, nullptr, scene->dev(), none, tp,
                           stack_cap, d_over);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e-3);
    }
    *seconds = best;
    *blocks_per_cu = occ;
    return 0;
}
template <int MODE, bool WIDE>
int probe_trace(const rt_scene *scene, const TraceParams &tp, int stack_cap, int *d_over, int cus, hipEvent_t e0, hipEvent_t e1,
                int minw, double *seconds, double *blocks_per_cu) {
    switch (minw) {
        case 8: return probe_trace_once<MODE, WIDE, 8>(scene, tp, stack_cap, d_over, cus, e0, e1, seconds, blocks_per_cu);
        case 6: return probe_trace_once<MODE, WIDE, 6>(scene, tp, stack_cap, d_over, cus, e0, e1, seconds, blocks_per_cu);
        case 5: return probe_trace_once<MODE, WIDE, 5>(scene, tp, stack_cap, d_over, cus, e0, e1, seconds, blocks_per_cu);
        default: return probe_trace_once<MODE, WIDE, 4>(scene, tp, stack_cap, d_over, cus, e0, e1, seconds, blocks_per_cu);
    }
}
}  // namespace


extern "C" {

int rt_measure_copy_bandwidth(int64_t bytes, int reps, double *out_bytes_per_s) {
    if (bytes < 1024 || reps < 1 || !out_bytes_per_s) return fail("rt_measure_copy_bandwidth: bad argument");
    size_t n4 = (size_t)bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    DevScope tmp;
    if (tmp.alloc(a, n4) || tmp.alloc(b, n4)) return 1;
    HIP_TRY(hipMemset(a, 1, n4 * 16));
    HIP_TRY(hipEventCreate(&tmp.e0));
    HIP_TRY(hipEventCreate(&tmp.e1));
    const hipEvent_t e0 = tmp.e0, e1 = tmp.e1;
    double best = 0.0;
    for (int r = 0; r < reps + 1; r++) {
        HIP_TRY(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(k_copy_f4, dim3(256 * 8), dim3(256), 0, nullptr, a, b, n4);
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms > 0.f) best = std::max(best, 2.0 * (double)n4 * 16.0 / (ms * 1e-3));
    }
    *out_bytes_per_s = best;
    return 0;
}

int rt_calibrate_valu(int waves_per_simd, int iters, double *out_lane_ops_per_s, double *out_wave_instr) {
    if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || !out_lane_ops_per_s) return fail("rt_calibrate_valu: bad argument");
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    // one 256-thread workgroup = one wave on each of a CU's four SIMDs
    const int blocks = cus * waves_per_simd;
    float *d_out = nullptr;
    DevScope tmp;
    if (tmp.alloc(d_out, (size_t)blocks * 256)) return 1;
    HIP_TRY(hipEventCreate(&tmp.e0));
    HIP_TRY(hipEventCreate(&tmp.e1));
    const hipEvent_t e0 = tmp.e0, e1 = tmp.e1;
    double best = 0.0;
    for (int r = 0; r < 7; r++) {  // (first launch untimed; clocks ramp: the best of six)
        HIP_TRY(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(k_valu_calibrate, dim3(blocks), dim3(256), 0, nullptr, d_out, iters, 1.f);
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        const double lane_ops = (double)blocks * 256.0 * 16.0 * (double)iters;
        if (r > 0 && ms > 0.f) best = std::max(best, lane_ops / (ms * 1e-3));
    }
    HIP_TRY(hipGetLastError());
    *out_lane_ops_per_s = best;
    if (out_wave_instr) *out_wave_instr = (double)blocks * 4.0 * 16.0 * (double)iters;  // v_fma_f32 wave-instructions per launch
    return 0;
}

int rt_calibrate_valu_packed(int waves_per_simd, int iters, int kind, double *out_lane_ops_per_s) {
    if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || kind < 1 || kind > 3 || !out_lane_ops_per_s)
        return fail("rt_calibrate_valu_packed: bad argument");
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int blocks = cus * waves_per_simd;
    float *d_out = nullptr;
    DevScope tmp;
    if (tmp.alloc(d_out, (size_t)blocks * 256)) return 1;
    HIP_TRY(hipEventCreate(&tmp.e0));
    HIP_TRY(hipEventCreate(&tmp.e1));
    double best = 0.0;
    for (int r = 0; r < 7; r++) {  // (first launch untimed; the best of six)
        HIP_TRY(hipEventRecord(tmp.e0, nullptr));
        if (kind == 1) hipLaunchKernelGGL(k_valu_calibrate_pk<1>, dim3(blocks), dim3(256), 0, nullptr, d_out, iters, 1.f);
        else if (kind == 2) hipLaunchKernelGGL(k_valu_calibrate_pk<2>, dim3(blocks), dim3(256), 0, nullptr, d_out, iters, 1.f);
        else hipLaunchKernelGGL(k_valu_calibrate_pk<3>, dim3(blocks), dim3(256), 0, nullptr, d_out, iters, 1.f);
        HIP_TRY(hipEventRecord(tmp.e1, nullptr));
        HIP_TRY(hipEventSynchronize(tmp.e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, tmp.e0, tmp.e1));
        const double lane_ops = (double)blocks * 256.0 * 16.0 * 2.0 * (double)iters;  // two lane-operations per lane and instruction
        if (r > 0 && ms > 0.f) best = std::max(best, lane_ops / (ms * 1e-3));
    }
    HIP_TRY(hipGetLastError());
    *out_lane_ops_per_s = best;
    return 0;
}

int rt_probe_issue(int kind, int waves_per_simd, int iters, double *out_seconds, double *out_wave_instr_per_wave) {
    if (kind < 0 || kind >= kProbeKinds || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || !out_seconds)
        return fail("rt_probe_issue: bad argument");
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int blocks = cus * waves_per_simd;
    float *d_out = nullptr;
    DevScope tmp;
    if (tmp.alloc(d_out, (size_t)blocks * 256)) return 1;
    HIP_TRY(hipEventCreate(&tmp.e0));
    HIP_TRY(hipEventCreate(&tmp.e1));
    double best = 1e30;
    for (int r = 0; r < 5; r++) {  // (first launch untimed; the best of four)
        HIP_TRY(hipEventRecord(tmp.e0, nullptr));
        switch (kind) {
#define RT_CASE(K) case K: hipLaunchKernelGGL(k_probe_issue<K>, dim3(blocks), dim3(256), 0, nullptr, d_out, iters, 1.f); break;
            RT_CASE(0) RT_CASE(1) RT_CASE(2) RT_CASE(3) RT_CASE(4) RT_CASE(5) RT_CASE(6) RT_CASE(7) RT_CASE(8) RT_CASE(9)
            RT_CASE(10) RT_CASE(11) RT_CASE(12) RT_CASE(13) RT_CASE(14) RT_CASE(15) RT_CASE(16) RT_CASE(17) RT_CASE(18) RT_CASE(19)
            RT_CASE(20) RT_CASE(21) RT_CASE(22) RT_CASE(23) RT_CASE(24) RT_CASE(25)
#undef RT_CASE
        }
        HIP_TRY(hipEventRecord(tmp.e1, nullptr));
        HIP_TRY(hipEventSynchronize(tmp.e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, tmp.e0, tmp.e1));
        if (r > 0 && ms > 0.f) best = std::min(best, (double)ms * 1e-3);
    }
    HIP_TRY(hipGetLastError());
    *out_seconds = best;
    if (out_wave_instr_per_wave)
        *out_wave_instr_per_wave = (kind >= 22 ? 64.0 : (kind == 11 || kind == 19) ? 32.0 : 16.0) * (double)iters;
    return 0;
}

int rt_split_probe(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples, int max_bounces,
                   uint64_t seed, int64_t target_rays, double *out, int n_out) {
    if (!scene || !camera || !out || n_out < RT_PROBE_COUNT) return fail("rt_split_probe: bad argument");
    if (width <= 0 || height <= 0 || num_samples <= 0 || max_bounces < 0 || target_rays < 1 || target_rays > (1LL << 30))
        return fail("rt_split_probe: bad dimensions");
    const long long cam_end = (long long)width * height * num_samples;
    if (cam_end + 13LL * kW >= (1LL << 31)) return fail("rt_split_probe: frame exceeds the int32 camera-ray range");
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != scene->device) return fail("rt_split_probe: scene was created on another device");
    if (int rc = ensure_origin_radius(scene, camera->lookfrom)) return rc;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    for (int k = 0; k < n_out; k++) out[k] = 0.0;
    const int n = kW;
    Context *cp = nullptr;
    if (get_context(n, 7, &cp)) return 1;  // (a context of its own: lane 7)
    Context &c = *cp;
    std::lock_guard<std::mutex> busy_lock(c.busy);
    double rng_seconds = 0.0;
    if (ensure_rng(c, seed, 0, nullptr, &rng_seconds)) return 1;
    DevScope pb;
    HIP_TRY(hipEventCreate(&pb.e0));
    HIP_TRY(hipEventCreate(&pb.e1));
    const unsigned cap = (unsigned)(target_rays + 2 * (long long)kW);
    float *c_o3, *c_d3, *c_tmax, *a_o3, *a_d3, *a_tmax, *rec, *outp, *fb, *o_t, *o_u, *o_v;
    int *a_excl, *o_i;
    unsigned *d_counts;
    if (pb.alloc(c_o3, 3 * (size_t)cap) || pb.alloc(c_d3, 3 * (size_t)cap) || pb.alloc(c_tmax, cap) || pb.alloc(a_o3, 3 * (size_t)cap) ||
        pb.alloc(a_d3, 3 * (size_t)cap) || pb.alloc(a_tmax, cap) || pb.alloc(a_excl, cap) || pb.alloc(rec, 3 * (size_t)kProbeIn * cap) ||
        pb.alloc(outp, (size_t)kProbeOut * cap) || pb.alloc(fb, 3 * (size_t)width * height) || pb.alloc(o_i, cap) || pb.alloc(o_t, cap) ||
        pb.alloc(o_u, cap) || pb.alloc(o_v, cap) || pb.alloc(d_counts, 8))
        return 1;
    HIP_TRY(hipMemset(d_counts, 0, sizeof(unsigned) * 8));
    HIP_TRY(hipMemset(fb, 0, sizeof(float) * 3 * (size_t)width * height));
    HIP_TRY(hipMemset(c.d_rows, 0, sizeof(DWaveRow) * (size_t)c.n_rows));
    {
        DCounters zero{};
        zero.last_live_round = -1;
        HIP_TRY(hipMemcpy(c.d_ctr, &zero, sizeof(DCounters), hipMemcpyHostToDevice));
    }
    DScene sc = scene->dev();
    Camera cam;
    memcpy(&cam, camera, sizeof(Camera));
    AdvanceParams ap{};
    ap.n = n;
    ap.slot_lo = 0;
    ap.width = width;
    ap.height = height;
    ap.spp = num_samples;
    ap.max_bounces = max_bounces;
    ap.cam_end = cam_end;
    ap.last_gen = (int)((cam_end + kW - 1) / kW) - 1;
    ap.batch_mask = 7;
    ap.w_over_spp = (kW % num_samples == 0) ? kW / num_samples : 0;
    ap.dpx = ap.w_over_spp % width;
    ap.dpy = (ap.w_over_spp > 0 && width < 32768 && height < 32768) ? ap.w_over_spp / width : -1;
    const bool lds_tables = scene->n_mats <= kLdsTable && scene->n_lights <= kLdsTable;
    const int stack_cap = lds_stack_cap(scene, kLdsStack);
    const size_t lds_bytes = sizeof(int) * (size_t)kBlock * (size_t)(stack_cap + 2);
    if (ensure_overflow(c.d_over, c.over_levels, scene->stack_bound - std::min(stack_cap, lds_stack_cap(scene, kPathsLdsStack)))) return 1;
    int occ_c = 0;
    if (scene->wide) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, k_trace<MODE_POOL, true>, kBlock, lds_bytes));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, k_trace<MODE_POOL, false>, kBlock, lds_bytes));
    const dim3 grid(grid_for(n)), block(kBlock), grid_trace(std::min(grid_for(n), std::max(1, cus * std::max(1, occ_c))));
    TraceParams tpp{};
    tpp.total = n;
    tpp.fb = fb;
    tpp.rows = c.d_rows;
    hipLaunchKernelGGL(k_pool_init, grid, block, 0, nullptr, c.pools, n, max_bounces);
    HIP_TRY(hipGetLastError());
    // ---- the round pipeline, with the dumps between its stages; every stage timed with events (synchronously: a probe)
    unsigned h_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double t_adv = 0.0, t_trace = 0.0, t_adv0 = 0.0;
    int rounds = 0;
    auto timed = [&](double &acc) -> int {
        HIP_TRY(hipEventRecord(pb.e1, nullptr));
        HIP_TRY(hipEventSynchronize(pb.e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pb.e0, pb.e1));
        acc += (double)ms * 1e-3;
        return 0;
    };
    while ((long long)h_counts[0] + h_counts[1] < target_rays && rounds < 4096) {
        ap.round = rounds;
        double t = 0.0;
        HIP_TRY(hipEventRecord(pb.e0, nullptr));
        RT_LAUNCH_ADVANCE(nullptr, fb);
        if (timed(t)) return 1;
        if (rounds == 0) t_adv0 = t;  // every slot generates: gen() alone
        else t_adv += t;
        hipLaunchKernelGGL(k_probe_dump_rays, grid, block, 0, nullptr, c.pools, n, c_o3, c_d3, c_tmax, a_o3, a_d3, a_tmax, a_excl, cap, d_counts);
        HIP_TRY(hipEventRecord(pb.e0, nullptr));
        RT_LAUNCH_TRACE(MODE_POOL, scene->wide, grid_trace, lds_bytes, nullptr, sc, c.pools, tpp, stack_cap, c.d_over);
        if (timed(t_trace)) return 1;
        hipLaunchKernelGGL(k_probe_dump_shades, grid, block, 0, nullptr, sc, c.pools, n, max_bounces, rec, cap, d_counts);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(h_counts, d_counts, sizeof(h_counts), hipMemcpyDeviceToHost));
        rounds++;
        if (h_counts[0] == 0) break;  // (frame exhausted)
    }
    c.rng_valid = false;  // the pools' RNG arrays have moved on
    const unsigned n_c = std::min(h_counts[0], cap), n_a = std::min(h_counts[1], cap);
    out[RT_PROBE_ROUNDS] = rounds;
    out[RT_PROBE_CLOSEST_RAYS] = n_c;
    out[RT_PROBE_ANY_RAYS] = n_a;
    out[RT_PROBE_S_ADVANCE_ROUND0] = t_adv0;
    out[RT_PROBE_S_ADVANCE] = t_adv;
    out[RT_PROBE_S_TRACE_POOL] = t_trace;
    // ---- trace only: the stage-level modes of k_trace on the dense ray arrays, at four register budgets
    {
        TraceParams tc{};
        tc.total = (int)n_c;
        tc.o3 = c_o3;
        tc.d3 = c_d3;
        tc.tmax = c_tmax;
        tc.order = scene->d_order;
        tc.out_i = o_i;
        tc.out_t = o_t;
        tc.out_u = o_u;
        tc.out_v = o_v;
        TraceParams ta{};
        ta.total = (int)n_a;
        ta.o3 = a_o3;
        ta.d3 = a_d3;
        ta.tmax = a_tmax;
        ta.excluded = a_excl;
        ta.out_i = o_i;
        const int budgets[4] = {8, 6, 5, 4};
        for (int v = 0; v < 4; v++) {
            double sc_s = 0.0, sa_s = 0.0, bc = 0.0, ba = 0.0;
            if (n_c > 0) {
                if (scene->wide ? probe_trace<MODE_TEST_CLOSEST, true>(scene, tc, stack_cap, c.d_over, cus, pb.e0, pb.e1, budgets[v], &sc_s, &bc)
                                : probe_trace<MODE_TEST_CLOSEST, false>(scene, tc, stack_cap, c.d_over, cus, pb.e0, pb.e1, budgets[v], &sc_s, &bc))
                    return 1;
            }
            if (n_a > 0) {
                if (scene->wide ? probe_trace<MODE_TEST_ANY, true>(scene, ta, stack_cap, c.d_over, cus, pb.e0, pb.e1, budgets[v], &sa_s, &ba)
                                : probe_trace<MODE_TEST_ANY, false>(scene, ta, stack_cap, c.d_over, cus, pb.e0, pb.e1, budgets[v], &sa_s, &ba))
                    return 1;
            }
            out[RT_PROBE_S_TRACE_CLOSEST + v] = sc_s;
            out[RT_PROBE_S_TRACE_ANY + v] = sa_s;
            out[RT_PROBE_TRACE_BLOCKS_PER_CU + v] = bc;
        }
    }
    // ---- shade only: one launch per material kind, every lane shading
    for (int mk = 0; mk < 3; mk++) {
        const unsigned cnt = std::min(h_counts[2 + mk], cap);
        out[RT_PROBE_SHADES + mk] = cnt;
        if (cnt == 0) continue;
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            HIP_TRY(hipEventRecord(pb.e0, nullptr));
            if (lds_tables)
                hipLaunchKernelGGL(k_probe_shade<true>, dim3((cnt + kBlock - 1) / kBlock), block, 0, nullptr, sc, cam, ap,
                                   rec + (size_t)mk * kProbeIn * cap, cap, cnt, outp, fb);
            else
                hipLaunchKernelGGL(k_probe_shade<false>, dim3((cnt + kBlock - 1) / kBlock), block, 0, nullptr, sc, cam, ap,
                                   rec + (size_t)mk * kProbeIn * cap, cap, cnt, outp, fb);
            HIP_TRY(hipGetLastError());
            double t = 0.0;
            if (timed(t)) return 1;
            best = std::min(best, t);
        }
        out[RT_PROBE_S_SHADE + mk] = best;
    }
    return 0;
}


}  // extern "C"
