/* rtcuda_amd_tools.h -- the LAB: measurement entry points of librtcuda_amd_tools.so (rtcuda_amd/csrc/rt_tools.hip).
 *
 * NOT part of the drop-in C-ABI (include/rtcuda_amd.h): nothing here replaces a reference interface -- the reference measures
 * nothing.  bench.py uses the two roofs (copy bandwidth, vector-ALU issue) for its roofline block; the scripts under tools/ use the rest.
 * The tools library is built from the product's own translation unit plus these tools, so the scene handle of
 * rt_split_probe must come from the tools library's own rt_scene_create. */
#ifndef RTCUDA_AMD_TOOLS_H
#define RTCUDA_AMD_TOOLS_H

#include <stdint.h>

#include "rtcuda_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measured device copy bandwidth in bytes/s (float4 copy of `bytes` bytes, best of `reps`):
 * the HBM roofline denominator SURVEY.md section 8d asks to be measured in the same run. */
int rt_measure_copy_bandwidth(int64_t bytes, int reps, double *out_bytes_per_s);

/* Measured vector-ALU roof in lane-operations/s: `waves_per_simd` waves on every SIMD of the chip, each issuing
 * iters x 16 independent v_fma_f32 (best of 6 timed launches).  The render kernels are bound by VALU issue, not
 * by HBM, so this -- next to the 157.3 TFLOP/s = 78.6 T lane-FMA/s spec -- is what bench.py's roofline divides by.
 * out_wave_instr (may be NULL): v_fma_f32 wave-instructions of one launch, for calibrating PMC counters.
 * No reference counterpart (the reference measures nothing). */
int rt_calibrate_valu(int waves_per_simd, int iters, double *out_lane_ops_per_s, double *out_wave_instr);

/* The same measurement with PACKED fp32 instructions (kind 1: v_pk_fma_f32, 2: v_pk_mul_f32, 3: v_pk_add_f32) on aligned
 * register pairs; lane-operations are counted as two per lane and instruction.  No reference counterpart. */
int rt_calibrate_valu_packed(int waves_per_simd, int iters, int kind, double *out_lane_ops_per_s);

/* How long ONE wave needs per instruction of a given kind, with `waves_per_simd` waves on every SIMD: out_seconds = best launch
 * time of a kernel in which every wave issues *out_wave_instr_per_wave instructions of kind (16 independent chains per lane
 * unless noted): 0 v_fma_f32 (VOP3), 1 v_fmac_f32, 2 v_mul_f32, 3 v_add_f32, 4 v_mov_b32, 5 v_xor_b32, 6 v_lshlrev_b32,
 * 7 v_max_f32, 8 v_rcp_f32, 9 v_sqrt_f32, 10 v_cndmask_b32, 11 v_mul_f32 + dependent v_add_f32, 12 ONE dependent chain of
 * v_fma_f32, 13 one dependent chain of v_mul_f32, 14 v_mul_f32 with a literal, 15 v_mul_f32 with an SGPR operand,
 * 16 v_fma_f32 with two SGPR operands, 17 v_cndmask_b32 with an SGPR-pair mask, 18 v_cmp_lt_f32 -> vcc, 19 v_cmp_lt_f32 +
 * dependent v_cndmask_b32, 20 v_bfi_b32, 21 v_and_b32, 22 - 25 three v_mul_f32 + one v_cndmask_b32 / v_max_f32 / v_mul_f32 /
 * v_mov_b32 per chain (an instruction's cost inside a mix).  No reference counterpart (tools/issue_probe.py). */
int rt_probe_issue(int kind, int waves_per_simd, int iters, double *out_seconds, double *out_wave_instr_per_wave);

/* Measurement tool for the design question "one persistent kernel, or the reference's stage split (render.cuh:428-449:
 * init/mat/gen kernels and ah/ch kernels with dense queues between them)?".  Runs the round-per-launch pipeline of the
 * frame from its start until `target_rays` rays have been traced, copying every round's rays (closest-hit and any-hit
 * apart) and every round's shading inputs (by material kind) into dense device arrays; then times, on those arrays,
 *   - the trace kernel alone (stage-level modes of the product's trace kernel) at 8 / 6 / 5 / 4 waves per SIMD,
 *   - the shading code alone (the product's init() + mat()), one material kind per launch, every lane shading,
 * each as the best of three launches (HIP events).  out[] is indexed by the RT_PROBE_* enum; times in seconds.
 * No reference counterpart. */
enum {
    RT_PROBE_ROUNDS = 0,          /* rounds of the pipeline that were run and dumped */
    RT_PROBE_CLOSEST_RAYS,        /* rays in the closest-hit array */
    RT_PROBE_ANY_RAYS,            /* rays in the any-hit array */
    RT_PROBE_S_ADVANCE_ROUND0,    /* k_advance of round 0: every slot runs gen() */
    RT_PROBE_S_ADVANCE,           /* k_advance, rounds 1.. (sum) */
    RT_PROBE_S_TRACE_POOL,        /* k_trace over the pools, all rounds (sum) */
    RT_PROBE_S_TRACE_CLOSEST,     /* [4]: closest-hit array at 8 / 6 / 5 / 4 waves per SIMD */
    RT_PROBE_S_TRACE_ANY = RT_PROBE_S_TRACE_CLOSEST + 4,            /* [4]: any-hit array */
    RT_PROBE_TRACE_BLOCKS_PER_CU = RT_PROBE_S_TRACE_ANY + 4,        /* [4]: resident 256-thread blocks per CU of each build */
    RT_PROBE_SHADES = RT_PROBE_TRACE_BLOCKS_PER_CU + 4,             /* [3]: shading records per kind (matte, mirror, glass) */
    RT_PROBE_S_SHADE = RT_PROBE_SHADES + 3,                         /* [3]: shading-only launch per kind */
    RT_PROBE_COUNT = RT_PROBE_S_SHADE + 3
};
int rt_split_probe(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
                   int max_bounces, uint64_t seed, int64_t target_rays, double *out, int n_out);

#ifdef __cplusplus
}
#endif
#endif /* RTCUDA_AMD_TOOLS_H */
