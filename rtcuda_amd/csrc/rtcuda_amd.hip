// rtcuda_amd.hip -- HIP kernels and the C-ABI of the MI355X-native render path (gfx950 only).
//
// The hot path of lashhw/rtcuda (render.cuh:61-457) re-designed for CDNA4:
//
//   * Path state lives in structure-of-arrays pools indexed by path SLOT (the reference keeps
//     28-byte AoS rays and pointer-chasing payloads: render.cuh:5-23).
//   * Slot s owns RNG stream s and serves exactly the camera rays c with c % W == s
//     (W = 1048576): in the reference every slot regenerates in lockstep (bounces is incremented
//     unconditionally, render.cuh:126, and reset only by gen, :268), so the rank of a slot in the
//     compacted gen queue is always the slot id itself.  A slot's history therefore never depends
//     on any other slot, and a slot may start its next camera ray as soon as its current path can
//     do nothing more.  That removes the reference's idle iterations (the
//     active fraction decays 100 % -> 2 % inside every 11-iteration generation) without changing
//     one random number, and makes the image invariant under any partition of the slots -- which
//     is how the work is sharded over GPUs.
//   * One global condition remains -- the host loop stops at the first iteration in which nothing
//     shades (render.cuh:436) -- and it can only bite in the final generation, which is therefore
//     run in lockstep (one init() per slot per round; the stop rule is evaluated on the device).
//
// Kernels:
//   k_paths      everything before the final generation, ONE persistent launch per frame.  A lane
//                owns a slot (then its next one); init+mat, gen, the shadow ray and the path ray are
//                PHASES of the lane; the wave issues, per iteration, the one block most of its lanes
//                wait for.  Rays, hit records and queues never leave registers / LDS; a camera
//                ray's contributions are summed in LDS and reach the framebuffer as one atomic triple;
//                traversal is speculative (a leaf reached inside a node block is set aside).
//   k_advance    init() + mat() + gen() for all slots of a round (render.cuh:84-275), state in the
//   k_trace      SoA pools; ch() + ah() of a round (render.cuh:278-328) with persistent waves, ballot +
//                mbcnt compaction into a per-wave LDS queue instead of flag arrays + CUB select
//                (render.cuh:348-364), while-while traversal, LDS stack.  Used for the lockstep final
//                generation, by the stage-level test entry points, and (RT_PERSISTENT=0) for whole frames.
//   advance_core / inner_step / tri_intersect / box_hit are the shared device functions: one copy of
//   the estimator and of the traversal for both pipelines.
//   * BVH: 64-byte node records with full-precision padded boxes -- a 4-wide node as two consecutive
//     records (default), or 2-wide nodes of one record (RT_BVH_WIDE=0) --
//     and 48-byte {p0,e1,e2,n} triangle records in leaf order (rt_bvh.h: SAH sweep + insertion-based
//     optimisation + collapse to 4-wide).
//   * Two results that depend, in the reference, on the shape of its own tree are defined by the triangle
//     list alone here: an accepted hit is never culled (conservative box test), and hits at exactly equal
//     t go to the larger caller index (closest_hit_wins).  Traversal ORDER therefore never matters.
//   * There are no host read-backs inside a frame (the reference does four blocking 4-byte read-backs per
//     iteration: render.cuh:433-434,444-445): the persistent kernel needs none, and the lockstep rounds of the
//     final generation carry their stop rule on the device (k_advance: `lock_shades`).
//
// No MFMA anywhere: there is no dense contraction on this path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/rtcuda_amd.h"
#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_ref_tree.h"

using namespace rt;

// ============================================================================ error handling
namespace {
thread_local std::string g_last_error;
std::string g_peer_log;  // rt_render_multi: what became of peer access, pair by pair (rt_peer_access_log)
int fail(const std::string &msg) {
    g_last_error = msg;
    return 1;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                        std::to_string(__LINE__) + ")");                                       \
    } while (0)

using rtbvh::knob;  // (experiment knobs are read only under RTCUDA_EXPERIMENTAL=1: rt_bvh.h)

constexpr int kW = RT_NUM_WORKING_PATHS;
constexpr uint32_t kFlagFixedFb = 0x200u;  // internal: d_sum points to int64 fixed-point sums
constexpr int kBlock = 256;       // 4 waves per workgroup
constexpr int kLdsStack = 16;          // traversal stack entries kept in LDS per lane (k_trace)
constexpr int kPathsLdsStack = 10;     // ... by k_paths: 10 + 1 + 26 rows = 37 KB per workgroup, four workgroups per CU; with 10 the
                                       // step without overflow handling (inner_step<WIDE, SHALLOW>) serves 95 % of the node steps
constexpr int kOverStride = 1 << 20;   // lanes of the overflow stack (>= lanes of the largest grid that traverses)
constexpr int kMaxStackBound = 160;    // deepest traversal stack a scene may need (3 per level + 1)
constexpr int kLockChunk = 16;         // lockstep rounds of the final generation enqueued between two looks at the stop rule's counters
}  // namespace

// ============================================================================ device structures
struct DScene {
    const float4 *nodes;   // 64-byte records (4 x float4): two per 4-wide node, one per 2-wide node (rt_bvh.h)
    const float4 *tris;    // 3 x float4 per triangle, leaf order
    const int2 *tri_info;  // leaf order: {material index, light index or -1}
    const float4 *tri_shade;  // leaf order: what mat() needs of a hit triangle besides the point -- the flipped unit
                              // normal -normalize(n) (render.cuh:153) and the packed ids (material | light + 1 << 16)
    const int *order;         // leaf order -> the caller's triangle index (closest-hit tie rule, test output)
    const Material *mats;
    const Light *lights;
    int num_lights;
    int num_mats;
    // shading tables in one block of dwords: [materials 5/each][lights 8/each][light triangle
    // records 12/each][per-light precomputed {1/area, unit normal} 4/each]
    const float *tables;
    int tab_dwords;
    // RT_FLAG_REFERENCE_WALK only (null otherwise; no other kernel reads them): the reference's own binary tree
    // (rt_ref_tree.h) as 32-byte nodes {bounds[6], count, link} = 2 x float4, and its primitive order mapped to this
    // scene's leaf-order triangle indices
    const float4 *ref_nodes;
    const int *ref_prims;
    int ref_n_prims;
    // default kernels (VERIFY; null with RT_FLAG_WATERTIGHT): leaf-order triangle index -> the node of its leaf in the
    // reference's tree, node -> parent node (root: -1), and whether that tree's root is a leaf (ref_visible)
    const int *ref_leaf_of;
    const int *ref_parent;
    int ref_root_leaf;
};
__device__ __host__ inline int tab_off_lights(int n_mats) { return 5 * n_mats; }
__device__ __host__ inline int tab_off_ltri(int n_mats, int n_lights) { return 5 * n_mats + 8 * n_lights; }
__device__ __host__ inline int tab_off_lpre(int n_mats, int n_lights) { return 5 * n_mats + 20 * n_lights; }

// Structure-of-arrays path state for the slots of one shard (n slots each)
// Structure-of-arrays path state for the slots of one shard: A_COUNT arrays of n dwords in ONE
// allocation, array k at base + k * n.  A kernel therefore carries one base pointer (2 SGPRs)
// instead of 36 array pointers -- the persistent kernel otherwise spends VGPRs and scratch on
// addresses.
//   ox..dz            current path ray
//   hit_info          -1 = miss, else material | (light index + 1) << 16
//   hpx..hpz          Triangle::p(u, v) of the last closest hit              (render.cuh:152)
//   hnx..hnz          -d_triangle->n.unit_vector()                           (render.cuh:153)
//   br, bg, bb        beta
//   bounces           as PathRayPayload::bounces; kDone / kParked are sentinels
//   pixel, gen        pixel of the current camera ray; index of the slot's NEXT generation
//   rd, r0..r4        XORWOW state
//   sox..slb, starget shadow ray of the slot for this round (stmax < 0: none) + radiance + excluded triangle
enum { A_OX, A_OY, A_OZ, A_DX, A_DY, A_DZ, A_HPX, A_HPY, A_HPZ, A_HNX, A_HNY, A_HNZ, A_BR, A_BG, A_BB, A_SOX, A_SOY, A_SOZ, A_SDX, A_SDY, A_SDZ, A_STMAX, A_SLR, A_SLG, A_SLB, A_HIT_INFO, A_BOUNCES, A_PIXEL, A_GEN, A_STARGET, A_RD, A_R0, A_R1, A_R2, A_R3, A_R4, A_COUNT };
struct DPools {
    float *base;
    int n;
    __device__ __forceinline__ float &ox(int i) const { return base[(unsigned)(A_OX * n + i)]; }
    __device__ __forceinline__ float &oy(int i) const { return base[(unsigned)(A_OY * n + i)]; }
    __device__ __forceinline__ float &oz(int i) const { return base[(unsigned)(A_OZ * n + i)]; }
    __device__ __forceinline__ float &dx(int i) const { return base[(unsigned)(A_DX * n + i)]; }
    __device__ __forceinline__ float &dy(int i) const { return base[(unsigned)(A_DY * n + i)]; }
    __device__ __forceinline__ float &dz(int i) const { return base[(unsigned)(A_DZ * n + i)]; }
    __device__ __forceinline__ float &hpx(int i) const { return base[(unsigned)(A_HPX * n + i)]; }
    __device__ __forceinline__ float &hpy(int i) const { return base[(unsigned)(A_HPY * n + i)]; }
    __device__ __forceinline__ float &hpz(int i) const { return base[(unsigned)(A_HPZ * n + i)]; }
    __device__ __forceinline__ float &hnx(int i) const { return base[(unsigned)(A_HNX * n + i)]; }
    __device__ __forceinline__ float &hny(int i) const { return base[(unsigned)(A_HNY * n + i)]; }
    __device__ __forceinline__ float &hnz(int i) const { return base[(unsigned)(A_HNZ * n + i)]; }
    __device__ __forceinline__ float &br(int i) const { return base[(unsigned)(A_BR * n + i)]; }
    __device__ __forceinline__ float &bg(int i) const { return base[(unsigned)(A_BG * n + i)]; }
    __device__ __forceinline__ float &bb(int i) const { return base[(unsigned)(A_BB * n + i)]; }
    __device__ __forceinline__ float &sox(int i) const { return base[(unsigned)(A_SOX * n + i)]; }
    __device__ __forceinline__ float &soy(int i) const { return base[(unsigned)(A_SOY * n + i)]; }
    __device__ __forceinline__ float &soz(int i) const { return base[(unsigned)(A_SOZ * n + i)]; }
    __device__ __forceinline__ float &sdx(int i) const { return base[(unsigned)(A_SDX * n + i)]; }
    __device__ __forceinline__ float &sdy(int i) const { return base[(unsigned)(A_SDY * n + i)]; }
    __device__ __forceinline__ float &sdz(int i) const { return base[(unsigned)(A_SDZ * n + i)]; }
    __device__ __forceinline__ float &stmax(int i) const { return base[(unsigned)(A_STMAX * n + i)]; }
    __device__ __forceinline__ float &slr(int i) const { return base[(unsigned)(A_SLR * n + i)]; }
    __device__ __forceinline__ float &slg(int i) const { return base[(unsigned)(A_SLG * n + i)]; }
    __device__ __forceinline__ float &slb(int i) const { return base[(unsigned)(A_SLB * n + i)]; }
    __device__ __forceinline__ int &hit_info(int i) const { return ((int *)base)[(unsigned)(A_HIT_INFO * n + i)]; }
    __device__ __forceinline__ int &bounces(int i) const { return ((int *)base)[(unsigned)(A_BOUNCES * n + i)]; }
    __device__ __forceinline__ int &pixel(int i) const { return ((int *)base)[(unsigned)(A_PIXEL * n + i)]; }
    __device__ __forceinline__ int &gen(int i) const { return ((int *)base)[(unsigned)(A_GEN * n + i)]; }
    __device__ __forceinline__ int &starget(int i) const { return ((int *)base)[(unsigned)(A_STARGET * n + i)]; }
    __device__ __forceinline__ uint32_t &rd(int i) const { return ((uint32_t *)base)[(unsigned)(A_RD * n + i)]; }
    __device__ __forceinline__ uint32_t &r0(int i) const { return ((uint32_t *)base)[(unsigned)(A_R0 * n + i)]; }
    __device__ __forceinline__ uint32_t &r1(int i) const { return ((uint32_t *)base)[(unsigned)(A_R1 * n + i)]; }
    __device__ __forceinline__ uint32_t &r2(int i) const { return ((uint32_t *)base)[(unsigned)(A_R2 * n + i)]; }
    __device__ __forceinline__ uint32_t &r3(int i) const { return ((uint32_t *)base)[(unsigned)(A_R3 * n + i)]; }
    __device__ __forceinline__ uint32_t &r4(int i) const { return ((uint32_t *)base)[(unsigned)(A_R4 * n + i)]; }
    // host-side address of array k
    float *array(int k) const { return base + (size_t)k * n; }
};

// Global words that need atomics / host polling.  Event counters are NOT here: they live in
// per-wave rows (DWaveRow) that only their owner wave updates, with plain loads and stores --
// 16384 waves hitting eight shared words with atomics every round was the single largest cost of
// the first version of k_advance.
struct DCounters {
    int last_live_round;         // highest batch-closing round in which some slot still traced a ray
    unsigned int unused0;        // (round 3: the lockstep rounds' shade count, now per round in Context::d_lock)
    unsigned int pad2[2];
    // VERIFY builds (rare events, global atomics): [0] accepted hits whose OWN box fails the reference's slab test,
    // [1] ... whose reference leaf box (or, for a ray with a -0.0 direction component, some ancestor box) fails it too =
    // hits the reference's walk loses, [2] closest hits with an exact tie at the final distance, [3] literal re-traces
    unsigned long long vstat[4];
};
enum { V_OWN_FAIL = 0, V_LOST = 1, V_TIE = 2, V_LITERAL = 3 };
enum { C_CAMERA = 0, C_SHADE, C_CLOSEST, C_ANY, C_EMIT, C_SHADOW_ADD, C_RR, C_UNUSED, C_COUNT };
struct DWaveRow {
    unsigned long long c[C_COUNT];  // one 64-byte line per wave
};

constexpr int kDone = -0x7fffffff;    // slot has no camera ray left
constexpr int kParked = -0x7ffffffe;  // slot waits for the lockstep rounds of the final generation

// ============================================================================ wave helpers
__device__ __forceinline__ unsigned lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ unsigned prefix_popc(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// Wave votes straight from the condition's compare (HIP's wave_ballot(int) first materialises the predicate as 0 / 1 in
// a VGPR and compares that with zero again: two more VALU instructions per vote, a dozen votes per scheduling decision).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int wave_count(bool p) { return (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }
__device__ __forceinline__ unsigned wave_index() { return (blockIdx.x * blockDim.x + threadIdx.x) >> 6; }
// lanes 0..7 of the wave add v[lane] to the wave's own row: one 64-byte load + one 64-byte store
__device__ __forceinline__ void row_add(DWaveRow *rows, const unsigned long long (&v)[C_COUNT]) {
    unsigned l = lane_id();
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < C_COUNT; k++) mine = (l == (unsigned)k) ? v[k] : mine;
    // no-return atomics on a line only this wave touches: fire-and-forget, no load round trip
    if (l < C_COUNT && mine != 0) atomicAdd(&rows[wave_index()].c[l], mine);
}

// ============================================================================ RNG init kernel
// curand_init(seed, slot, 0) (render.cuh:68-73): the seed-scrambled state advanced by slot * 2^67
// draws.  The 2^67-draw jump is the GF(2)-linear map J on the 160 state bits; jump_pow holds
// J^(2^k), k = 0..19, as 160 rows x 5 words each (row b = image of basis bit b), so J^slot is at
// most 20 mat-vecs selected by the bits of the slot id.
__global__ void k_rng_init(DPools p, int n, int slot_lo, Rng seed_state, const uint32_t *__restrict__ jump_pow) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t slot = (uint32_t)(slot_lo + i);
    uint32_t v[5] = {seed_state.v0, seed_state.v1, seed_state.v2, seed_state.v3, seed_state.v4};
    for (int k = 0; k < 20; k++) {
        if (!((slot >> k) & 1u)) continue;
        const uint32_t *m = jump_pow + (size_t)k * 160 * 5;
        uint32_t r[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 5; w++) {
            uint32_t bits = v[w];
            for (int b = 0; b < 32; b++) {
                uint32_t sel = 0u - ((bits >> b) & 1u);
                const uint32_t *row = m + (w * 32 + b) * 5;
                r[0] ^= row[0] & sel;
                r[1] ^= row[1] & sel;
                r[2] ^= row[2] & sel;
                r[3] ^= row[3] & sel;
                r[4] ^= row[4] & sel;
            }
        }
#pragma unroll
        for (int w = 0; w < 5; w++) v[w] = r[w];
    }
    p.rd(i) = seed_state.d;
    p.r0(i) = v[0];
    p.r1(i) = v[1];
    p.r2(i) = v[2];
    p.r3(i) = v[3];
    p.r4(i) = v[4];
}

// init_path_ray_payload (render.cuh:75-82): every slot starts "finished" so the first round
// routes it to gen.  (The reference stores INT_MAX and relies on INT_MAX+1 wrapping; any value
// >= max_bounces has the same effect on the first init().)
__global__ void k_pool_init(DPools p, int n, int max_bounces) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    p.hit_info(i) = -1;
    p.bounces(i) = max_bounces;
    p.gen(i) = 0;
    p.pixel(i) = 0;
}

// ============================================================================ k_advance
struct AdvanceParams {
    int n;        // slots in this shard
    int slot_lo;  // global id of local slot 0
    int width, height, spp, max_bounces;
    long long cam_end;  // width*height*spp
    int round;
    int batch_mask;      // rounds with (round & batch_mask) == batch_mask close a host-polled batch
    int last_gen;        // index of the final camera-ray generation
    int lockstep;        // != 0: final generation, one init() per slot per round (literal reference schedule); 1 + the index of
                         // the lockstep round (0 = the round that generates)
    int fb_fixed;        // framebuffer holds 64-bit fixed-point sums (see deposit())
    int w_over_spp;      // W / spp when spp divides W (then pixel = gen * w_over_spp + slot / spp: no 64-bit divide), else 0
    int dpx, dpy;        // w_over_spp = dpy * width + dpx: how a slot's pixel moves per generation; dpy < 0: not usable
    // RT_FLAG_RNG_PER_SAMPLE: every camera ray starts its own stream, keyed by its GLOBAL id = local id * key_mul + key_add
    // (rank `key_add` of `key_mul` renders the frame at spp / key_mul with the full slot pool); no slot parks
    int per_sample, key_mul, key_add;
    uint32_t seed_lo, seed_hi;
};

constexpr int kLdsTable = 64;                   // materials / lights staged in LDS per workgroup
constexpr int kTabDwordsMax = kLdsTable * 29;   // 5 + 8 + 12 + 4 dwords per (material, light)

// Per-light values that depend on the light triangle only, computed once per scene on the device
// with the same operations mat() would redo per shade: 1 / Triangle::area() (triangle.cuh:79,84-86)
// and d_triangle->n.unit_vector() (light.cuh:46).
// Per-triangle shading record, computed once per scene with the operations mat() would redo at every shade
// (isect_unit_n = -d_triangle->n.unit_vector(), render.cuh:153; vec3.cuh:131-134).
__global__ void k_build_tri_shade(const float4 *__restrict__ tris, const int2 *__restrict__ tri_info, int n,
                                  float4 *__restrict__ out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Tri tr = load_tri(tris, k);
    V3 un = neg(unit(tr.n));
    int2 ml = tri_info[k];
    out[k] = make_float4(un.x, un.y, un.z, __int_as_float((ml.x & 0xffff) | ((ml.y + 1) << 16)));
}

__global__ void k_build_tables(const Material *mats, int n_mats, const Light *lights, int n_lights,
                               const float4 *tris, float *tab) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_mats) {
        const float *src = (const float *)&mats[t];
        for (int k = 0; k < 5; k++) tab[5 * t + k] = src[k];
    }
    if (t < n_lights) {
        const float *src = (const float *)&lights[t];
        float *dl = tab + tab_off_lights(n_mats) + 8 * t;
        for (int k = 0; k < 8; k++) dl[k] = src[k];
        float *dt = tab + tab_off_ltri(n_mats, n_lights) + 12 * t;
        float *dp = tab + tab_off_lpre(n_mats, n_lights) + 4 * t;
        Light l = lights[t];
        if (l.type == 1) {
            Tri lt = load_tri(tris, l.tri);
            const float *q = (const float *)(tris + 3 * (size_t)l.tri);
            for (int k = 0; k < 12; k++) dt[k] = q[k];
            V3 un = unit(lt.n);
            dp[0] = 1.f / tri_area(lt);
            dp[1] = un.x;
            dp[2] = un.y;
            dp[3] = un.z;
        } else {
            for (int k = 0; k < 12; k++) dt[k] = 0.f;
            dp[0] = dp[1] = dp[2] = dp[3] = 0.f;
        }
    }
}

__device__ __forceinline__ Material tab_material(const float *tab, int i) {
    const float *q = tab + 5 * i;
    Material m;
    m.ax = q[0];
    m.ay = q[1];
    m.az = q[2];
    m.ior = q[3];
    m.type = __float_as_int(q[4]);
    return m;
}
__device__ __forceinline__ Light tab_light(const float *tab, int n_mats, int i) {
    const float *q = tab + tab_off_lights(n_mats) + 8 * i;
    Light l;
    l.type = __float_as_int(q[0]);
    l.px = q[1];
    l.py = q[2];
    l.pz = q[3];
    l.tri = __float_as_int(q[4]);
    l.lx = q[5];
    l.ly = q[6];
    l.lz = q[7];
    return l;
}

// ---------------------------------------------------------------------------------------------
// advance_core: init() + mat() + gen() for ONE slot (render.cuh:84-275), on register state.
// Shared by k_advance (state loaded from / stored to the pools) and k_paths (state lives in
// registers for the whole frame).
// Framebuffer deposit.  Default: three float atomics, as the reference's Vec3::atomic_add
// (vec3.cuh:149-153) -- the summation order, and with it the last bits of a pixel, vary from run to
// run.  Fixed mode (RT_FLAG_DETERMINISTIC / rt_render_shard_fixed): the buffer holds 64-bit
// fixed-point sums (scale 2^30); integer adds commute, so the image is bit-reproducible and the sum of
// the shards of a multi-GPU render is EXACTLY the single-GPU sum.  Non-finite contributions (none
// occur in any test scene) are dropped and magnitudes are clamped to 2^31 in that mode.
constexpr float kFixedScale = 1073741824.f;  // 2^30
__device__ __forceinline__ long long to_fixed(float x) {
    if (!(fabsf(x) <= 2147483648.f)) x = (x == x) ? copysignf(2147483648.f, x) : 0.f;
    return __float2ll_rn(x * kFixedScale);
}
__device__ __forceinline__ void deposit(float *__restrict__ fb, int fixed, int pixel, float r, float g, float b) {
    const unsigned k = (unsigned)(3 * pixel);
    if (fixed) {
        unsigned long long *f = (unsigned long long *)fb;
        atomicAdd(&f[k + 0], (unsigned long long)to_fixed(r));
        atomicAdd(&f[k + 1], (unsigned long long)to_fixed(g));
        atomicAdd(&f[k + 2], (unsigned long long)to_fixed(b));
    } else {
        atomicAdd(&fb[k + 0], r);
        atomicAdd(&fb[k + 1], g);
        atomicAdd(&fb[k + 2], b);
    }
}

// Per-lane sample accumulator of k_paths (3 floats in the lane's LDS column, element k at acc[k * kBlock]): the
// contributions of ONE camera ray -- bounce-0 emission, then the unoccluded shadow rays in path order -- are summed here
// and reach the framebuffer as one atomic triple when the slot starts its next camera ray (the reference issues one
// atomic triple per contribution, in an order that differs from run to run: vec3.cuh:149-153).  A float atomic is a
// fabric transaction whose acknowledgement a wave's later loads queue behind (vmcnt is in order), so the frame has
// 0.5 G of them instead of 1.0 G and they sit in the GEN block instead of the traversal loop: +4 %.
__device__ __forceinline__ void acc_add(float *acc, float r, float g, float b) {
    acc[0 * kBlock] += r;
    acc[1 * kBlock] += g;
    acc[2 * kBlock] += b;
}
__device__ __forceinline__ void acc_flush(float *acc, float *__restrict__ fb, int fixed, int pixel) {
    const float r = acc[0 * kBlock], g = acc[1 * kBlock], b = acc[2 * kBlock];
    if (r != 0.f || g != 0.f || b != 0.f) {  // (a NaN contribution compares unequal to 0: it is deposited)
        deposit(fb, fixed, pixel, r, g, b);
        acc[0 * kBlock] = 0.f;
        acc[1 * kBlock] = 0.f;
        acc[2 * kBlock] = 0.f;
    }
}

struct SlotState {
    int bounces, hit_info, pixel, gen;
    Rng rs;
    V3 beta, wo, isect_p, isect_n;
};
struct AdvanceOut {
    bool did_gen, did_shade, has_shadow, did_emit, new_ray;
    bool wants_gen;             // advance_core<DEFER_GEN = true> only: the slot's next step is gen()
    int rr_draws;
    V3 ray_o, ray_d;            // next path ray (valid when new_ray)
    V3 s_o, s_d, s_L;           // shadow ray + radiance to deposit if unoccluded (valid when has_shadow)
    float s_tmax;
    int s_target;
};

// gen() (render.cuh:250-275) for one slot.  Camera ray id = generation * W + slot (see file header).  Leaves
// st.bounces = kDone (no camera ray left) / kParked (the final generation runs in lockstep) or a new path.
// `pxy` (optional): the slot's previous pixel as (x | y << 16), or -1.  A slot's pixel index grows by W / spp per
// generation, so with it the pixel coordinates follow by an add and a carry instead of two integer divisions.
// NEVER_LOCKSTEP: the caller (k_paths) only ever runs with ap.lockstep == 0 -- known at compile time there, a value read
// from LDS (and so a divergent branch with its merges, as far as the compiler can tell) otherwise.
// `cid_given` >= 0 (per-sample streams on the persistent kernel only): the camera ray is not the slot's next one but the one
// the wave drew from the frame's counter -- with a stream of its own per camera ray, any lane can take any camera ray.
template <bool NEVER_LOCKSTEP = false>
__device__ __forceinline__ void gen_core(const Camera &cam, const AdvanceParams &ap, int slot_global, SlotState &st,
                                         AdvanceOut &out, int *pxy = nullptr, long long cid_given = -1) {
    const bool lockstep = NEVER_LOCKSTEP ? false : (ap.lockstep != 0);
    long long cid = cid_given >= 0 ? cid_given : (long long)st.gen * kW + slot_global;
    if (cid >= ap.cam_end) {
        st.bounces = kDone;
        return;
    }
    if (!lockstep && st.gen == ap.last_gen) {
        st.bounces = kParked;
        return;
    }
    st.gen = st.gen + 1;
    // pixel = camera_ray_id / spp (render.cuh:254-256).  cid = gen * W + slot, so when spp divides W the quotient
    // splits exactly into two 32-bit terms; the general case keeps the 64-bit division.
    int px, py;
    if (cid_given >= 0) {
        st.pixel = (int)((unsigned)cid / (unsigned)ap.spp);  // (camera-ray ids stay below 2^31: rt_render_shard checks)
        py = (int)((unsigned)st.pixel / (unsigned)ap.width);
        px = st.pixel - py * ap.width;
    } else if (pxy && ap.dpy >= 0 && *pxy >= 0) {
        px = (*pxy & 0xffff) + ap.dpx;
        py = (*pxy >> 16) + ap.dpy;
        if (px >= ap.width) {
            px -= ap.width;
            py++;
        }
        st.pixel = py * ap.width + px;
    } else {
        if (ap.w_over_spp) st.pixel = (st.gen - 1) * ap.w_over_spp + (int)((unsigned)slot_global / (unsigned)ap.spp);  // (gen already counts this ray)
        else st.pixel = (int)(cid / ap.spp);
        py = (int)((unsigned)st.pixel / (unsigned)ap.width);  // (both non-negative: the unsigned divide is the cheaper one)
        px = st.pixel - py * ap.width;
    }
    if (pxy) *pxy = px | (py << 16);
    if (ap.per_sample) st.rs = rng_sample_stream(ap.seed_lo, ap.seed_hi, (unsigned long long)cid * (unsigned)ap.key_mul + (unsigned)ap.key_add);
    float jx = rng_uniform(st.rs);  // x first, then y (Appendix A.7)
    float jy = rng_uniform(st.rs);
    camera_get_ray(cam, (px + jx) / ap.width, (py + jy) / ap.height, out.ray_o, out.ray_d);
    out.new_ray = true;
    st.bounces = 0;
    st.beta = mk(1.f, 1.f, 1.f);
    out.did_gen = true;
}

// `acc` (USE_ACC, k_paths only): the lane's sample accumulator; otherwise contributions go straight into the framebuffer.
template <bool DEFER_GEN, bool NEVER_LOCKSTEP = false, bool USE_ACC = false>
__device__ __forceinline__ void advance_core(const DScene &sc, const float *tab, const Camera &cam,
                                             const AdvanceParams &ap, int slot_global, SlotState &st, AdvanceOut &out,
                                             float *__restrict__ fb, float *acc = nullptr) {
    const bool lockstep = NEVER_LOCKSTEP ? false : (ap.lockstep != 0);
    const int off_ltri = tab_off_ltri(sc.num_mats, sc.num_lights);
    const int off_lpre = tab_off_lpre(sc.num_mats, sc.num_lights);
    out.did_gen = out.did_shade = out.has_shadow = out.did_emit = out.new_ray = out.wants_gen = false;
    out.rr_draws = 0;
    const bool hit = st.hit_info >= 0;
    const int light_of_hit = hit ? ((st.hit_info >> 16) & 0xffff) - 1 : -1;
    // Emulate consecutive init() calls (render.cuh:84-137) until one of them ends in mat() or gen(): a slot whose
    // path missed idles (no RNG use) until bounces reaches max_bounces; a slot that Russian roulette "killed" is
    // re-rolled by every following init() (Appendix A.1) -- beta, and with it the kill probability, does not change
    // along such a chain, so the chain is a tight loop of draws.  `lockstep`: exactly one init() per call.
    RT_MARK("adv.init");
    if (st.bounces == 0 && hit && light_of_hit >= 0) {  // :98-103 emission only at bounce 0
        Light l = tab_light(tab, sc.num_mats, light_of_hit);
        if (USE_ACC) acc_add(acc, l.lx, l.ly, l.lz);  // (k_paths; a compile-time choice: `if (acc)` is a per-lane pointer test)
        else deposit(fb, ap.fb_fixed, st.pixel, l.lx, l.ly, l.lz);
        out.did_emit = true;
    }
    const bool cont = st.bounces < ap.max_bounces;  // :109
    if (cont && hit) {
        bool shade = true;
        if (st.bounces > kRrStart && max3(st.beta) < kRrThreshold) {  // :112-124
            const float pt = fmaxf(0.05f, 1 - max3(st.beta));
            shade = false;
            while (true) {
                out.rr_draws++;
                const bool kill = rng_uniform(st.rs) < pt;
                st.bounces = st.bounces + 1;  // :126
                if (!kill) {
                    st.beta = divf(st.beta, 1 - pt);
                    shade = true;
                    break;
                }
                if (lockstep || !(st.bounces < ap.max_bounces)) break;
            }
        } else {
            st.bounces = st.bounces + 1;  // :126
        }
        if (shade) out.did_shade = true;
        else if (lockstep) return;  // killed this round; the next round rolls again
    } else {
        if (cont && lockstep) {  // a miss idles: nothing but the counter moves (Appendix A.2)
            st.bounces = st.bounces + 1;
            return;
        }
    }
    if (!out.did_shade) {
        // ---- gen() :250-275 (the init() that finds no bounce left; its own increment of `bounces` is overwritten)
        if (DEFER_GEN) {  // k_paths: camera rays are generated by the (much shorter) GEN block
            out.wants_gen = true;
            return;
        }
        if (USE_ACC) acc_flush(acc, fb, ap.fb_fixed, st.pixel);  // the camera ray that ends here: its sum -> its pixel
        gen_core<NEVER_LOCKSTEP>(cam, ap, slot_global, st, out);
        return;
    }
    // ---- mat() :139-248
    RT_MARK("adv.mat.sample_f");
    Material m = tab_material(tab, st.hit_info & 0xffff);
    V3 multiplier = scale(st.beta, (float)sc.num_lights);  // taken BEFORE the beta update (:150)
    int again_draws;
    {
        V3 n = st.isect_n, wi;
        float pdf;
        V3 f = mat_sample_f(m, st.wo, st.rs, n, wi, pdf, again_draws);
        out.ray_o = offset_ray_origin(st.isect_p, n);
        out.ray_d = wi;
        out.new_ray = true;
        st.beta = mul(st.beta, divf(scale(f, dot(wi, n)), pdf));  // :166
    }
    RT_MARK("adv.mat.light_sample");
    if (sc.num_lights > 0) {
        int light_idx = min((int)(rng_uniform(st.rs) * sc.num_lights), sc.num_lights - 1);  // :178
        Light light = tab_light(tab, sc.num_mats, light_idx);
        V3 wi, Li;
        float lt, lpdf;
        // Light::sample_Li light.cuh:29-48
        if (light.type == 0) {
            wi = sub(mk(light.px, light.py, light.pz), st.isect_p);
            lt = len(wi);
            Li = divf(mk(light.lx, light.ly, light.lz), lt * lt);
            wi = divf(wi, lt);
            lpdf = 1.f;
        } else {
            const float *q = tab + off_ltri + 12 * light_idx;
            const float *pre = tab + off_lpre + 4 * light_idx;
            Tri lt_tri;
            lt_tri.p0 = mk(q[0], q[1], q[2]);
            lt_tri.e1 = mk(q[3], q[4], q[5]);
            lt_tri.e2 = mk(q[6], q[7], q[8]);
            lt_tri.n = mk(q[9], q[10], q[11]);
            lpdf = pre[0];                        // 1 / area
            V3 lun = mk(pre[1], pre[2], pre[3]);   // unit normal of the light triangle
            float a = sqrtf(rng_uniform(st.rs));   // Triangle::sample_p triangle.cuh:78-82
            float u2 = rng_uniform(st.rs);
            V3 tp = tri_point(lt_tri, 1 - a, u2 * a);
            wi = sub(tp, st.isect_p);
            lt = len(wi);
            wi = divf(wi, lt);
            Li = mk(light.lx, light.ly, light.lz);
            lpdf *= len2(sub(tp, st.isect_p)) / fabsf(dot(lun, wi));
        }
        RT_MARK("adv.mat.nee");
        V3 n = dot(st.isect_n, wi) > 0.f ? st.isect_n : neg(st.isect_n);  // :187
        V3 f;
        float spdf;
        if (mat_get_f(m, st.wo, wi, n, f, spdf)) {
            f = scale(f, dot(wi, n));
            out.s_o = offset_ray_origin(st.isect_p, n);
            out.s_d = wi;
            out.s_tmax = lt;
            out.s_target = light.type == 1 ? light.tri : -1;
            if (light.type == 0) {
                out.s_L = divf(mul(mul(multiplier, f), Li), lpdf);  // :199
            } else {
                float weight = power_heuristic(lpdf, spdf);  // :201 (int-truncating)
                out.s_L = divf(scale(mul(mul(multiplier, f), Li), weight), lpdf);  // :202
            }
            out.has_shadow = true;
        }
        // "sample BSDF with MIS" block :213-245: its ray cannot contribute; keep its draws
        RT_MARK("adv.mat.burn");
        if (light.type != 0) {  // (the first call already knows how many: see mat_sample_f)
            if (again_draws >= 1) rng_next(st.rs);
            if (again_draws >= 2) rng_next(st.rs);
        }
    }
    RT_MARK("adv.mat.end");
}

// k_advance: one slot per thread, state in the pools.
// Every per-slot input is indexed by the slot id, so all of a lane's loads are issued together
// (one memory round trip); the small shared tables are staged in LDS with one more independent
// round trip; the hit record written by k_trace<MODE_POOL> already carries the shading point,
// the flipped unit normal and the material / light ids, so no triangle is gathered here.  The
// kernel has no atomics on shared words and one barrier (the table staging): the shadow ray goes
// to the slot's own record, event counts go to the wave's own counter row.
// SORT (material-sorted shading; opt-in with RT_SORT_SHADE=1): which slot a thread serves is decided per
// workgroup by the MATERIAL the slot is about to shade with.  The 256 slots of the block are partitioned -- ballot + mbcnt
// ranks per wave, wave offsets through 16 LDS counters -- into [matte | mirror | glass | nothing to shade] and thread t takes
// the t-th slot of that order, so a wave runs one branch of Material::sample_f (material.cuh:60-109) instead of all three
// (the reference shades in compacted-queue order, whatever the material: render.cuh:139-145).  A slot's computation does
// not depend on the thread that runs it, and every per-slot array is indexed by the slot: results are unchanged (a GPU
// test holds the fixed-point sums equal).  Measured on C2's whole frame through the round pipeline (RT_PERSISTENT=0):
// k_advance takes 87 ms sorted against 70 ms in slot order (frame 535 vs 517 ms, profiles/r04_experiments.md) -- the kernel
// streams 36 arrays of slot state and is bound by that traffic; the sort costs a dependent load phase and two barriers in
// front of it and scatters the accesses inside the block's window, while the divergence it removes (three short branches of
// sample_f) was not what the kernel waited for.  So slot order stays the default, here as in the persistent kernel.
template <bool LDS_TABLES, bool SORT = false>
__global__ void __launch_bounds__(kBlock)
k_advance(DScene sc, DPools p, Camera cam, AdvanceParams ap, float *__restrict__ fb, DCounters *__restrict__ ctr,
          DWaveRow *__restrict__ rows, unsigned int *__restrict__ lock_shades) {
    // Lockstep rounds (final generation): the reference's host loop ends at the first iteration in which nothing shades
    // (render.cuh:436).  The rounds are all enqueued at once; lock_shades[j] counts the mat() events of round j, and a round
    // finds out on the device whether the render ended before it: round j >= 2 does nothing if round j - 1 shaded nothing
    // (round 0 only generates; once a round is skipped it counts nothing, so every later one is skipped too).  No host poll.
    if (ap.lockstep >= 3 && lock_shades[ap.lockstep - 2] == 0u) return;
    __shared__ float s_tab[LDS_TABLES ? kTabDwordsMax : 1];
    __shared__ int s_perm[SORT ? kBlock : 1];
    __shared__ int s_count[SORT ? 16 : 1];  // [kind][wave of the block]

    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (SORT) {
        int kind = 3;  // nothing to shade: gen(), an idle or finished slot, a thread past the end
        if (i < ap.n) {
            const int b = p.bounces(i), hi = p.hit_info(i);
            if (b != kDone && b != kParked && hi >= 0 && b < ap.max_bounces)  // init() will route it to mat() (render.cuh:109,128-130)
                kind = min(max(__float_as_int(sc.tables[5 * (hi & 0xffff) + 4]), 0), 2);
        }
        const unsigned wave = threadIdx.x >> 6;
        unsigned rank = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = wave_ballot(kind == k);
            if (kind == k) rank = prefix_popc(m);
            if (lane_id() == 0) s_count[4 * k + wave] = (int)__popcll(m);
        }
        __syncthreads();
        int base = 0;
        for (int q = 0; q < 16; q++)  // everything of a smaller kind, and of this kind in the waves before this one
            base += (q < 4 * kind + (int)wave) ? s_count[q] : 0;
        s_perm[base + (int)rank] = (int)threadIdx.x;
        __syncthreads();
        i = blockIdx.x * blockDim.x + s_perm[threadIdx.x];
    }
    const bool in_range = i < ap.n;
    // ---- issue all per-slot loads up front
    SlotState st;
    st.bounces = kDone;
    st.hit_info = -1;
    st.pixel = 0;
    st.gen = 0;
    st.rs = Rng{0, 0, 0, 0, 0, 0};
    st.beta = st.wo = st.isect_p = st.isect_n = mk(0, 0, 0);
    if (in_range) {
        st.bounces = p.bounces(i);
        st.hit_info = p.hit_info(i);
        st.pixel = p.pixel(i);
        st.gen = p.gen(i);
        st.rs = Rng{p.rd(i), p.r0(i), p.r1(i), p.r2(i), p.r3(i), p.r4(i)};
        st.beta = mk(p.br(i), p.bg(i), p.bb(i));
        st.wo = mk(p.dx(i), p.dy(i), p.dz(i));
        st.isect_p = mk(p.hpx(i), p.hpy(i), p.hpz(i));
        st.isect_n = mk(p.hnx(i), p.hny(i), p.hnz(i));
    }
    const float *tab = sc.tables;
    if (LDS_TABLES) {
        for (int k = threadIdx.x; k < sc.tab_dwords; k += kBlock) s_tab[k] = sc.tables[k];
        __syncthreads();
        tab = s_tab;
    }
    // Final generation: the reference stops the whole render at the first iteration in which no slot
    // shades (render.cuh:436), which can cut off slots that Russian roulette would have revived
    // later.  That is a global condition, so the last generation runs in lockstep: slots that finish
    // generation last_gen - 1 park, and once all are parked the rounds run one init() per slot each (all enqueued at once:
    // see the top of this kernel for how a round knows that the render ended before it).
    if (ap.lockstep && st.bounces == kParked) {
        st.bounces = ap.max_bounces;  // routes the slot to gen() below
        st.hit_info = -1;
    }
    const bool alive = st.bounces != kDone && st.bounces != kParked;
    AdvanceOut out;
    out.did_gen = out.did_shade = out.has_shadow = out.did_emit = out.new_ray = false;
    out.rr_draws = 0;
    if (alive) {
        const int gen_before = st.gen;
        advance_core<false>(sc, tab, cam, ap, ap.slot_lo + i, st, out, fb);
        if (out.new_ray) {
            p.ox(i) = out.ray_o.x;
            p.oy(i) = out.ray_o.y;
            p.oz(i) = out.ray_o.z;
            p.dx(i) = out.ray_d.x;
            p.dy(i) = out.ray_d.y;
            p.dz(i) = out.ray_d.z;
        }
        if (st.gen != gen_before) {
            p.gen(i) = st.gen;
            p.pixel(i) = st.pixel;
        }
        if (out.has_shadow) {
            p.sox(i) = out.s_o.x;
            p.soy(i) = out.s_o.y;
            p.soz(i) = out.s_o.z;
            p.sdx(i) = out.s_d.x;
            p.sdy(i) = out.s_d.y;
            p.sdz(i) = out.s_d.z;
            p.slr(i) = out.s_L.x;
            p.slg(i) = out.s_L.y;
            p.slb(i) = out.s_L.z;
            p.starget(i) = out.s_target;
            p.stmax(i) = out.s_tmax;
        }
        p.br(i) = st.beta.x;
        p.bg(i) = st.beta.y;
        p.bb(i) = st.beta.z;
        p.bounces(i) = st.bounces;
        p.rd(i) = st.rs.d;
        p.r0(i) = st.rs.v0;
        p.r1(i) = st.rs.v1;
        p.r2(i) = st.rs.v2;
        p.r3(i) = st.rs.v3;
        p.r4(i) = st.rs.v4;
    }
    if (in_range && !out.has_shadow) p.stmax(i) = -1.f;  // no shadow ray from this slot this round

    // ---- event counters: this wave's own row
    unsigned long long traced = wave_ballot(out.did_gen || out.did_shade);
    int rr_tot = out.rr_draws;
    if (wave_ballot(out.rr_draws != 0)) {
        for (int off = 32; off > 0; off >>= 1) rr_tot += __shfl_xor(rr_tot, off);
    } else {
        rr_tot = 0;
    }
    unsigned long long v[C_COUNT] = {(unsigned long long)wave_count((out.did_gen)),
                                     (unsigned long long)wave_count((out.did_shade)),
                                     (unsigned long long)__popcll(traced),
                                     (unsigned long long)wave_count((out.has_shadow)),
                                     (unsigned long long)wave_count((out.did_emit)),
                                     0ull,
                                     (unsigned long long)rr_tot,
                                     0ull};
    row_add(rows, v);
    // liveness is monotone (a finished slot never restarts), so the host only needs it for the
    // round that closes a batch: one plain store per live wave in 1 round out of 8
    if ((ap.round & ap.batch_mask) == ap.batch_mask && traced != 0 && lane_id() == 0) ctr->last_live_round = ap.round;
    if (ap.lockstep) {
        unsigned long long sm = wave_ballot(out.did_shade);
        if (sm != 0 && lane_id() == 0) atomicAdd(&lock_shades[ap.lockstep - 1], (unsigned)__popcll(sm));
    }
}

// ============================================================================ traversal
// One wave-wide traversal engine serves the four trace entry points (closest-hit over the path
// pools = ch(), render.cuh:297-328; any-hit over the shadow queue = ah(), :278-294; and the two
// stage-level test hooks), so the parity tests exercise exactly the code the renderer runs.
//
// Structure (wave64, persistent):
//   * every wave owns 64 lanes = 64 rays in flight and keeps pulling ray indices from a global
//     head counter in chunks of kChunk (one atomic per 256 rays); finished lanes are finalised and
//     re-filled together once fewer than kRefillAt lanes are still traversing, so the wave does not
//     idle on its longest ray;
//   * "while-while": all lanes first step through inner pair records until each holds a leaf (or
//     is finished), then all lanes test their leaf's triangles -- node steps run beside node steps
//     and triangle tests beside triangle tests instead of serialising per lane;
//   * the traversal stack is a column of LDS per lane (replaces device_stack.cuh's int[29] in
//     scratch memory); entries are inner pair indices (>= 0) or leaf references (< 0).
//
// The box test only culls: it is conservative (boxes padded by the builder, exit distance widened
// by 8 ulp) and may use any arithmetic.  The triangle test is the reference's, bit for bit.
struct RayPrep {
    V3 o, d, inv;
};
__device__ __forceinline__ V3 inv_dir(V3 d) {
    // aabb_intersector.cuh:17-19 clamps |d| away from 0 the same way before inverting.  The reciprocal itself is the
    // hardware's v_rcp_f32 (1 ulp) rather than an IEEE division (11 instructions each, three per ray): 1 / d only feeds
    // the box test, which only culls -- box_hit / inner_step widen the exit distance by 8 ulps, which covers the 1 ulp
    // per axis this costs on top of the rounding of the slab arithmetic (the builder pads every box by 2 ulps)
    float ix = __builtin_amdgcn_rcpf((fabsf(d.x) < kFltEps) ? copysignf(kFltEps, d.x) : d.x);
    float iy = __builtin_amdgcn_rcpf((fabsf(d.y) < kFltEps) ? copysignf(kFltEps, d.y) : d.y);
    float iz = __builtin_amdgcn_rcpf((fabsf(d.z) < kFltEps) ? copysignf(kFltEps, d.z) : d.z);
    return mk(ix, iy, iz);
}
__device__ __forceinline__ bool box_hit(V3 o, V3 inv, float lox, float loy, float loz, float hix, float hiy,
                                        float hiz, float tmax, float &entry) {
    float ax = (lox - o.x) * inv.x, bx = (hix - o.x) * inv.x;
    float ay = (loy - o.y) * inv.y, by = (hiy - o.y) * inv.y;
    float az = (loz - o.z) * inv.z, bz = (hiz - o.z) * inv.z;
    float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    entry = t_in;
    t_out = t_out * 1.000001f;
    return t_in <= t_out && t_out >= 0.f && t_in <= tmax * 1.000001f;  // (tmax widened like t_out: see inner_step)
}

typedef float v2f __attribute__((ext_vector_type(2)));  // packed fp32 (v_pk_*_f32 on gfx950)
constexpr int kEntryDone = (int)0x80000000;  // "nothing left to visit" marker for a lane (== rtbvh::kNoChild)
// Traversal stack: the first `cap` entries of a lane live in its LDS column, deeper ones (rare: the
// bound is 3 per tree level, the typical depth under 10) in a per-lane column of a global overflow
// buffer, so LDS use -- and with it occupancy -- is set by the common case, not the worst case.
__device__ __forceinline__ void stack_push(int *lds_col, int *over_col, int &sp, int cap, int v) {
    if (sp < cap) lds_col[sp * kBlock] = v;
    else over_col[(size_t)(sp - cap) * kOverStride] = v;
    sp++;
}
// Pushes without divergent branches: a value is ALWAYS stored one above the current top of the LDS part and the stack
// pointer moves only if the push is meant (what lies above the top is never read).  The LDS part has one row more than
// `cap` (callers allocate cap + 1 rows), which takes the stores of lanes whose LDS part is full; only such lanes,
// rarely, branch -- once per node step -- to the global overflow column.  Written as `if (push) ...`, each of the up
// to 3 pushes of a 4-wide node step cost an exec-mask save / restore pair, two jumps and a 64-bit overflow address:
// a third of the instructions of the step.
__device__ __forceinline__ void push_if(int *lds_col, int *over_col, int &sp, int cap, int v, bool push) {
    lds_col[min(sp, cap) * kBlock] = v;
    if (push && sp >= cap) {
        over_col[(size_t)(sp - cap) * kOverStride] = v;
        __asm__ volatile("" ::: "memory");
    }
    sp += push ? 1 : 0;
}
__device__ __forceinline__ void push_if4(int *lds_col, int *over_col, int &sp, int cap, int v0, bool p0, int v1, bool p1,
                                         int v2, bool p2, int v3, bool p3) {
    const int s0 = sp, s1 = s0 + (p0 ? 1 : 0), s2 = s1 + (p1 ? 1 : 0), s3 = s2 + (p2 ? 1 : 0), s4 = s3 + (p3 ? 1 : 0);
    lds_col[min(s0, cap) * kBlock] = v0;
    lds_col[min(s1, cap) * kBlock] = v1;
    lds_col[min(s2, cap) * kBlock] = v2;
    lds_col[min(s3, cap) * kBlock] = v3;
    if (s4 > cap) {  // rare: some of this lane's pushes belong in the overflow column
        if (p0 && s0 >= cap) over_col[(size_t)(s0 - cap) * kOverStride] = v0;
        if (p1 && s1 >= cap) over_col[(size_t)(s1 - cap) * kOverStride] = v1;
        if (p2 && s2 >= cap) over_col[(size_t)(s2 - cap) * kOverStride] = v2;
        if (p3 && s3 >= cap) over_col[(size_t)(s3 - cap) * kOverStride] = v3;
        __asm__ volatile("" ::: "memory");
    }
    sp = s4;
}
__device__ __forceinline__ int stack_pop(int *lds_col, int *over_col, int &sp, int cap) {
    sp--;
    // always read the LDS column (clamped) and patch from the overflow only when needed: written as a
    // select of two pointers, the compiler merges the paths into one FLAT load, which is slower
    int v = lds_col[min(sp, cap - 1) * kBlock];
    if (sp >= cap) {
        v = over_col[(size_t)(sp - cap) * kOverStride];
        __asm__ volatile("" ::: "memory");  // keeps this a branch: merged, the two loads become one FLAT load behind
                                            // a dozen instructions of 64-bit address selection, on every pop
    }
    return v;
}
// Closest hit among EQUAL distances.  The reference accepts `t <= tmax` (triangle.cuh:49), so of two triangles hit at
// exactly the same t (a shared edge) the one its BVH walk tests LAST wins -- a property of the
// reference's tree that no other tree can reproduce.  Here the tie goes to the triangle with the larger index in the
// CALLER's order, whatever the tree: the result is a function of the ray and the triangle list alone (the oracle's
// watertight mode applies the same rule; ties are ~1 in 10^7 rays).  `tri` / `tmax`: best hit so far.
__device__ __forceinline__ bool closest_hit_wins(const DScene &sc, float t, float tmax, int k, int tri) {
    if (t == tmax && tri >= 0) return sc.order[(unsigned)k] > sc.order[(unsigned)tri];
    return true;
}
constexpr int kRefillAt = 40;                // finalise + refill once <= this many lanes still traverse
__device__ __forceinline__ int leaf_ref(int first, int count) { return ~((first << 3) | count); }

// One node step of a lane whose current entry is an inner record (cur >= 0): test the children,
// continue with the nearest one that the ray may enter, push the others (far first).
// `top` / `top_n`: the first top_n records (the top of the tree, breadth-first: rt_bvh.h) may be staged
// in LDS by the caller; nullptr / 0 otherwise.
// SHALLOW (4-wide nodes, k_paths): the caller has established -- with one wave vote -- that every lane taking this step has at
// most stack_cap - 3 entries, so neither the pop nor the up to three pushes of the step can leave the LDS part of the stack:
// no clamps, no overflow branches (each of which costs the wave an exec-mask save / restore pair and a jump whether or not a
// lane takes it; the general step has four such rare regions).  95 % of the node steps of C2 qualify at stack_cap = 10.
template <bool WIDE, bool SHALLOW = false>
__device__ __forceinline__ void inner_step(const DScene &sc, V3 o, V3 inv, float tmax, int &cur, int &sp, int *stack,
                                           int *over, int stack_cap, const float4 *top = nullptr, int top_n = 0) {
    // (2-wide records) the top of the LDS part of the stack, in case this step ends in a pop: see below
    const int spec_top = SHALLOW ? stack[max(sp - 1, 0) * kBlock] : stack[max(min(sp - 1, stack_cap - 1), 0) * kBlock];
    // 2-wide: one 64-byte record, q0..q3.  4-wide: the node's 128 bytes are laid out BY PLANE (upload_node_records): per axis a
    // 16-byte word with the four children's lower bounds and one with their upper bounds, then the four links.  Which of the two
    // is the NEAR plane of an axis depends on the sign of 1 / d alone, so each lane fetches near and far planes by address
    // (word index 2 * axis + sign, and the other one) and the slab test needs no min / max per axis: 24 instructions fewer per
    // node step than sorting each pair of distances.  n*: near planes, f*: far planes, q3: links.
    float4 q0, q1, q2, q3, nx, ny, nz, fx, fy, fz;
    if (WIDE) {
        const unsigned bx = (__float_as_uint(inv.x) >> 27) & 16u, by = (__float_as_uint(inv.y) >> 27) & 16u,
                       bz = (__float_as_uint(inv.z) >> 27) & 16u;  // 16 where 1 / d is negative: the upper bound is the near one
        if (top_n > 0 && cur < top_n) {
            const char *q = (const char *)(top + 4 * cur);
            nx = *(const float4 *)(q + bx);
            fx = *(const float4 *)(q + (bx ^ 16u));
            ny = *(const float4 *)(q + 32 + by);
            fy = *(const float4 *)(q + 32 + (by ^ 16u));
            nz = *(const float4 *)(q + 64 + bz);
            fz = *(const float4 *)(q + 64 + (bz ^ 16u));
            q3 = *(const float4 *)(q + 96);
            __asm__ volatile("" ::: "memory");  // (keeps the two branches apart: see below)
        } else {
            const char *q = (const char *)sc.nodes;
            const unsigned base = (unsigned)cur << 6;
            nx = *(const float4 *)(q + (base | bx));
            fx = *(const float4 *)(q + ((base | bx) ^ 16u));
            ny = *(const float4 *)(q + ((base | by) + 32u));
            fy = *(const float4 *)(q + (((base | by) ^ 16u) + 32u));
            nz = *(const float4 *)(q + ((base | bz) + 64u));
            fz = *(const float4 *)(q + (((base | bz) ^ 16u) + 64u));
            q3 = *(const float4 *)(q + (base + 96u));
        }
        q0 = q1 = q2 = q3;  // (unused in this form)
    } else if (top_n > 0 && cur < top_n) {
        const float4 *q = top + 4 * cur;
        q0 = q[0];
        q1 = q[1];
        q2 = q[2];
        q3 = q[3];
        // keeps the two branches apart: merged into a select of pointers they become FLAT loads, which go
        // through the texture addresser like any global load and make the LDS copy pointless
        __asm__ volatile("" ::: "memory");
        nx = ny = nz = fx = fy = fz = q0;
    } else {
        const float4 *q = (const float4 *)((const char *)sc.nodes + ((unsigned)cur << 6));
        q0 = q[0];
        q1 = q[1];
        q2 = q[2];
        q3 = q[3];
        nx = ny = nz = fx = fy = fz = q0;
    }
    if (!WIDE) {
        // 2-wide record: two exact boxes, near child first, far child onto the stack.  The bounds of the two
        // children are interleaved (rt_scene_create), so the 12 subtractions and 12 multiplications of the
        // slab test are 6 + 6 packed operations; each component is rounded exactly as in box_hit.
        int cl = __float_as_int(q3.x), cr = __float_as_int(q3.y);
        const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
        const v2f ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
        v2f ax = v2f{q0.x, q0.y} - ox, ay = v2f{q0.z, q0.w} - oy, az = v2f{q1.x, q1.y} - oz;
        v2f bx = v2f{q1.z, q1.w} - ox, by = v2f{q2.x, q2.y} - oy, bz = v2f{q2.z, q2.w} - oz;
        ax = ax * ix; ay = ay * iy; az = az * iz;
        bx = bx * ix; by = by * iy; bz = bz * iz;
        const float el = fmaxf(fmaxf(fminf(ax.x, bx.x), fminf(ay.x, by.x)), fminf(az.x, bz.x));
        const float er = fmaxf(fmaxf(fminf(ax.y, bx.y), fminf(ay.y, by.y)), fminf(az.y, bz.y));
        v2f t_out = {fminf(fminf(fmaxf(ax.x, bx.x), fmaxf(ay.x, by.x)), fmaxf(az.x, bz.x)),
                     fminf(fminf(fmaxf(ax.y, bx.y), fmaxf(ay.y, by.y)), fmaxf(az.y, bz.y))};
        t_out = t_out * v2f{1.000001f, 1.000001f};
        // (tmax is widened like the exit distance: the entry distance carries the same few ulps of rounding, and a
        // shadow ray that ends ON a triangle coplanar with an occluder -- light quads -- has entry = t = tmax to the
        // last bit; unwidened, the full-size audit of the sixteen-light scene lost 1 occluder in 9.8e8 shadow rays)
        const float tmax_w = tmax * 1.000001f;
        bool hl = el <= t_out.x && t_out.x >= 0.f && el <= tmax_w && cl != kEntryDone;
        bool hr = er <= t_out.y && t_out.y >= 0.f && er <= tmax_w && cr != kEntryDone;
        // What comes next, with as little divergent control flow as the three outcomes allow (every divergent branch
        // costs the wave an exec-mask save / restore pair and a jump, a dozen scalar instructions per step before):
        //   one child entered  -> it becomes the cursor;
        //   both               -> the nearer one, the farther one onto the stack (the only branch left, a single store);
        //   none               -> the top of the stack, read speculatively BEFORE the slab arithmetic (`spec_top`), so
        //                         that the LDS latency of a pop is never on the critical path of a step.
        const bool both = hl && hr, none = !(hl || hr);
        const bool left_first = !(el > er);
        int popped = sp > 0 ? spec_top : kEntryDone;
        if (none && sp > stack_cap) {  // rare: the entry lives in the global overflow part
            popped = over[(size_t)(sp - 1 - stack_cap) * kOverStride];
            __asm__ volatile("" ::: "memory");
        }
        const int entered = (hl && (!hr || left_first)) ? cl : cr;
        cur = none ? popped : entered;
        sp -= (none && sp > 0) ? 1 : 0;
        push_if(stack, over, sp, stack_cap, left_first ? cr : cl, both);
    }
    if (WIDE) {
        // 4-wide node = two pair-style records (children 0, 1 | children 2, 3) with full-precision boxes: half the
        // dependent fetches of the 2-wide walk for the same box arithmetic.  The nearest child the ray may enter becomes
        // the cursor, the others go onto the stack in record order (measured on the CPU walk: sorting them as well
        // saves 0.3 % of the steps), nothing entered -> the speculative top of the stack.
        const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y), c2 = __float_as_int(q3.z), c3 = __float_as_int(q3.w);
        const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
        const v2f ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
        // (clamped to a finite value: an absent child has an all-+inf box -- rt_bvh.h -- whose entry distance is +inf or
        // whose exit distance is -inf whatever the ray, so the one comparison below rejects it without a look at its link)
        const float tmax_w = fminf(tmax * 1.000001f, kFltMax);
        float e[4];
        bool h[4];
        // entered <=> entry <= exit, exit >= 0, entry <= tmax: max(entry, 0) <= min(exit, tmax) -- one comparison per child
        // instead of three and their scalar ANDs (tmax >= 0 always)
        // plane distance = b * (1 / d) + s, s = -o * (1 / d): ONE packed fma per pair of planes where (b - o) * (1 / d) takes
        // two instructions.  s is rounded on its own, which moves the planes of an axis by up to 2^-24 |o| as the ray sees
        // them: the records are padded for that (rt_bvh.h, pad_quads_for_origins; ensure_origin_radius on the host).
        // Near and far planes were picked by the sign of 1 / d when they were fetched: monotone rounding makes the near
        // plane's distance the smaller of the two, the very value min() would pick.
        const v2f sx = {-o.x * inv.x, -o.x * inv.x}, sy = {-o.y * inv.y, -o.y * inv.y}, sz = {-o.z * inv.z, -o.z * inv.z};
        (void)ox; (void)oy; (void)oz; (void)q0; (void)q1; (void)q2;
#define RT_SLAB(b, i, s_) __builtin_elementwise_fma((b), (i), (s_))
        {
            const v2f tnx = RT_SLAB((v2f{nx.x, nx.y}), ix, sx), tny = RT_SLAB((v2f{ny.x, ny.y}), iy, sy), tnz = RT_SLAB((v2f{nz.x, nz.y}), iz, sz);
            const v2f tfx = RT_SLAB((v2f{fx.x, fx.y}), ix, sx), tfy = RT_SLAB((v2f{fy.x, fy.y}), iy, sy), tfz = RT_SLAB((v2f{fz.x, fz.y}), iz, sz);
            e[0] = fmaxf(fmaxf(tnx.x, tny.x), tnz.x);
            e[1] = fmaxf(fmaxf(tnx.y, tny.y), tnz.y);
            v2f t_out = {fminf(fminf(tfx.x, tfy.x), tfz.x), fminf(fminf(tfx.y, tfy.y), tfz.y)};
            t_out = t_out * v2f{1.000001f, 1.000001f};
            h[0] = fmaxf(e[0], 0.f) <= fminf(t_out.x, tmax_w);
            h[1] = fmaxf(e[1], 0.f) <= fminf(t_out.y, tmax_w);
        }
        {
            const v2f tnx = RT_SLAB((v2f{nx.z, nx.w}), ix, sx), tny = RT_SLAB((v2f{ny.z, ny.w}), iy, sy), tnz = RT_SLAB((v2f{nz.z, nz.w}), iz, sz);
            const v2f tfx = RT_SLAB((v2f{fx.z, fx.w}), ix, sx), tfy = RT_SLAB((v2f{fy.z, fy.w}), iy, sy), tfz = RT_SLAB((v2f{fz.z, fz.w}), iz, sz);
            e[2] = fmaxf(fmaxf(tnx.x, tny.x), tnz.x);
            e[3] = fmaxf(fmaxf(tnx.y, tny.y), tnz.y);
            v2f t_out = {fminf(fminf(tfx.x, tfy.x), tfz.x), fminf(fminf(tfx.y, tfy.y), tfz.y)};
            t_out = t_out * v2f{1.000001f, 1.000001f};
            h[2] = fmaxf(e[2], 0.f) <= fminf(t_out.x, tmax_w);
            h[3] = fmaxf(e[3], 0.f) <= fminf(t_out.y, tmax_w);
        }
#undef RT_SLAB
        // nearest entered child (a child that is not entered counts as infinitely far)
        const float f0 = h[0] ? e[0] : kFltMax, f1 = h[1] ? e[1] : kFltMax, f2 = h[2] ? e[2] : kFltMax, f3 = h[3] ? e[3] : kFltMax;
        const bool a01 = !(f0 > f1), a23 = !(f2 > f3);       // winner of each record (ties: the lower index)
        const float g01 = a01 ? f0 : f1, g23 = a23 ? f2 : f3;
        const bool first = !(g01 > g23);
        const int near_k = first ? (a01 ? 0 : 1) : (a23 ? 2 : 3);
        const int near_link = first ? (a01 ? c0 : c1) : (a23 ? c2 : c3);
        const bool any_hit = h[0] || h[1] || h[2] || h[3];
        int spec = spec_top;
        __asm__ volatile("" : "+v"(spec));  // the read stays where it was issued: the compiler otherwise sinks it into a branch
        int popped = sp > 0 ? spec : kEntryDone;
        if (!SHALLOW && !any_hit && sp > stack_cap) {
            popped = over[(size_t)(sp - 1 - stack_cap) * kOverStride];
            __asm__ volatile("" ::: "memory");
        }
        cur = any_hit ? near_link : popped;
        sp -= (!any_hit && sp > 0) ? 1 : 0;
        if (SHALLOW) {  // (every value is stored one above the running top; the top moves only if the push is meant)
            const bool p0 = h[0] && near_k != 0, p1 = h[1] && near_k != 1, p2 = h[2] && near_k != 2, p3 = h[3] && near_k != 3;
            const int s0 = sp, s1 = s0 + (p0 ? 1 : 0), s2 = s1 + (p1 ? 1 : 0), s3 = s2 + (p2 ? 1 : 0);
            stack[s0 * kBlock] = c0;
            stack[s1 * kBlock] = c1;
            stack[s2 * kBlock] = c2;
            stack[s3 * kBlock] = c3;
            sp = s3 + (p3 ? 1 : 0);
        } else {
            push_if4(stack, over, sp, stack_cap, c0, h[0] && near_k != 0, c1, h[1] && near_k != 1, c2, h[2] && near_k != 2, c3,
                     h[3] && near_k != 3);
        }
    }
}

// ============================================================================ RT_FLAG_REFERENCE_WALK
// The reference's own traversal over its own tree (rt_ref_tree.h), decision for decision -- opt-in, never timed:
//   * AABBIntersector (aabb_intersector.cuh:14-36): octant from the sign of d, 1 / d as an IEEE division with |d|
//     clamped away from 0, scaled origin (-o) * (1 / d); per slab inv * bound + scaled_origin as a separately rounded
//     multiplication and addition (this file is built with -ffp-contract=off); hit iff entry <= exit -- on the exact,
//     unpadded boxes, with no clamp to [0, tmax].  This is the test that loses about one accepted hit in 10^7 rays;
//   * Bvh::traverse (bvh.cuh:251-303 / :306-357): the two children of a node are tested left then right, a leaf
//     child is intersected on the spot (left leaf before right leaf), of two inner children the one with the smaller
//     entry distance is descended first (ties: the left one) and the other one's children index is pushed;
//   * intersect_leaf (:222-236 / :239-248): triangles in the reference's primitive order; closest hit accepts
//     t <= tmax, so the LATER tested of two hits at equal t wins (triangle.cuh:49); any hit returns at the first
//     accepted triangle that is not the excluded one.
// A lane runs its whole ray here in one go (a plain per-lane loop with a private stack of 32 entries -- the
// reference's DeviceStack has 29, device_stack.cuh:4-11, for a tree of depth <= 30): no speculation, no reordering.
// `tri`: best hit so far / excluded triangle, as everywhere else (leaf-order index); ANY sets hu = 1 when occluded.
struct RefSlab {
    bool nx, ny, nz;  // octant: direction component negative
    V3 inv, so;
};
__device__ inline RefSlab ref_slab(V3 o, V3 d) {
    RefSlab s;
    s.nx = d.x < 0;
    s.ny = d.y < 0;
    s.nz = d.z < 0;
    s.inv = mk(1.f / ((fabsf(d.x) < kFltEps) ? copysignf(kFltEps, d.x) : d.x),
               1.f / ((fabsf(d.y) < kFltEps) ? copysignf(kFltEps, d.y) : d.y),
               1.f / ((fabsf(d.z) < kFltEps) ? copysignf(kFltEps, d.z) : d.z));
    s.so = mul(neg(o), s.inv);
    return s;
}
// node = {xmin, xmax, ymin, ymax | zmin, zmax, count, link}
__device__ inline bool ref_box(const RefSlab &s, float4 n0, float4 n1, float &entry) {
    const float ex = s.inv.x * (s.nx ? n0.y : n0.x) + s.so.x;
    const float ey = s.inv.y * (s.ny ? n0.w : n0.z) + s.so.y;
    const float ez = s.inv.z * (s.nz ? n1.y : n1.x) + s.so.z;
    entry = fmaxf(ex, fmaxf(ey, ez));
    const float xx = s.inv.x * (s.nx ? n0.x : n0.y) + s.so.x;
    const float xy = s.inv.y * (s.ny ? n0.z : n0.w) + s.so.y;
    const float xz = s.inv.z * (s.nz ? n1.x : n1.y) + s.so.z;
    const float exit = fminf(xx, fminf(xy, xz));
    return entry <= exit;
}
// `stack` / `over` / `cap`: the lane's own traversal stack (LDS column + global overflow column, stack_push / stack_pop) --
// free whenever this runs, since the lane's ray through the product's tree has ended or never started.  (Round 4 kept 32
// entries in a private array: the compiler promoted it to 32 VGPRs indexed through select chains -- 227 VGPRs unconstrained,
// 53 spilled at the 4-wave budget.)
template <bool ANY>
__device__ inline void reference_walk(const DScene &sc, V3 o, V3 d, float &tmax, int &tri, float &hu, float &hv, int *stack,
                                      int *over, int cap) {
    if (sc.ref_n_prims <= 0) return;
    const float4 *__restrict__ nodes = sc.ref_nodes;
    // true: the ray is finished (an occluder was found)
    auto leaf = [&](int first, int count) -> bool {
#pragma nounroll
        for (int i = first; i < first + count; i++) {
            const int k = sc.ref_prims[i];
            const Tri tr = load_tri(sc.tris, k);
            float t, u, v;
            if (tri_intersect(tr, o, d, tmax, t, u, v)) {
                if (ANY) {
                    if (k != tri) {
                        hu = 1.f;
                        return true;
                    }
                } else {
                    tmax = t;
                    hu = u;
                    hv = v;
                    tri = k;
                }
            }
        }
        return false;
    };
    {
        const float4 r1 = nodes[1];
        if (__float_as_int(r1.z) > 0) {  // the root is a leaf (:252 / :307)
            leaf(__float_as_int(r1.w), __float_as_int(r1.z));
            return;
        }
    }
    const RefSlab s = ref_slab(o, d);
    int sp = 0;
    int left = __float_as_int(nodes[1].w);
    // (a walk over a validated tree of n nodes ends after at most n / 2 pairs; the bound is a guard, not a schedule)
#pragma nounroll
    for (int guard = 0; guard < (1 << 24); guard++) {
        const float4 a0 = nodes[2 * left], a1 = nodes[2 * left + 1], b0 = nodes[2 * left + 2], b1 = nodes[2 * left + 3];
        const int lcount = __float_as_int(a1.z), llink = __float_as_int(a1.w);
        const int rcount = __float_as_int(b1.z), rlink = __float_as_int(b1.w);
        float el, er;
        bool go_l = ref_box(s, a0, a1, el);
        if (go_l && lcount > 0) {
            if (leaf(llink, lcount)) return;
            go_l = false;
        }
        bool go_r = ref_box(s, b0, b1, er);
        if (go_r && rcount > 0) {
            if (leaf(rlink, rcount)) return;
            go_r = false;
        }
        if (go_l && go_r) {
            const bool right_first = el > er;
            stack_push(stack, over, sp, cap, right_first ? llink : rlink);
            left = right_first ? rlink : llink;
        } else if (go_l) {
            left = llink;
        } else if (go_r) {
            left = rlink;
        } else {
            if (sp == 0) break;
            left = stack_pop(stack, over, sp, cap);
        }
    }
}

// ============================================================================ VERIFY: the reference's decisions on the product's walk
// What the reference's walk can SEE is a function of the ray alone: its box test does not look at tmax
// (aabb_intersector.cuh:35), so a leaf is reached iff every box on the way down to it passes `entry <= exit`, whatever has
// been hit before.  Its closest hit is therefore the nearest accepted triangle AMONG THE VISIBLE ONES (ties: the one its
// walk tests last, triangle.cuh:49), and a shadow ray is occluded iff a VISIBLE accepted triangle other than the target
// exists -- definitions that any search order over any acceleration structure can evaluate.  And visibility is cheap:
//   * the boxes along a root-to-leaf path are nested exactly (a node's box is the min / max of its triangles' boxes:
//     bvh.cuh:57-61,150-160); fp32 rounding is monotone, so each slab term inv * bound + scaled_origin is a monotone
//     function of the bound, non-decreasing for inv > 0 and non-increasing for inv < 0; with the octant chosen by the
//     sign of d (aabb_intersector.cuh:14-16) the near bound of a parent gives an entry distance <= its child's and the
//     far bound an exit distance >= its child's.  Hence: IF THE LEAF'S BOX PASSES, EVERY ANCESTOR'S PASSES -- a triangle
//     is visible iff its LEAF's box passes the reference's test (nothing is assumed about the size of any rounding error);
//   * the triangle's own box (triangle.cuh:22-37) lies inside its leaf's, so a pass on the own box -- computed from the
//     record that is in registers anyway -- is a pass on the leaf's: the common case costs no memory access.  Only when
//     the own box fails (flat boxes of axis-aligned triangles hit on their rim: ~1 hit in 10^7) is the leaf's box fetched;
//   * the one case in which octant and sign of 1 / d disagree is a direction component of exactly -0.0 (d < 0 is false,
//     1 / copysign(eps, -0.0) is negative): the nesting argument does not hold then and the ancestors are tested one by
//     one through the parent links.
// So the default kernels keep their own tree, node format, speculation and scheduling and still return the reference's
// answers: a shadow ray's accepted hit only counts if its triangle is visible (k_trace: the walk goes on past an unseen
// occluder; k_paths: the ray ends at its first occluder, and if the reference cannot see that one -- ~1 in 10^7 -- the
// literal walk decides), and a finished path ray's closest hit T is checked once, in the block that shades it anyway: T
// visible and no exact tie at the final distance  =>  T is the reference's closest hit (T is the nearest accepted triangle
// of ALL, so also of the visible ones).  The rest -- T invisible (the nearest VISIBLE hit is needed) or a tie (the
// reference's test order decides) -- is ~2 rays in 10^7 and is re-traced by reference_walk behind a rare branch.  tests/test_traversal_audit.py replays > 4 * 10^7 rays of literal oracle renders through the CPU twin of
// exactly this procedure (rt_host_check.cpp): equal on every ray; the GPU suite holds whole frames to the LITERAL
// oracle's fixed-point image bit for bit.
__device__ __forceinline__ bool neg_zero3(V3 d) {
    return __float_as_uint(d.x) == 0x80000000u || __float_as_uint(d.y) == 0x80000000u || __float_as_uint(d.z) == 0x80000000u;
}
__device__ __forceinline__ bool ref_visible(const DScene &sc, V3 o, V3 d, const Tri &tr, int k,
                                            unsigned long long *__restrict__ vstat) {
    // the reference's slab setup (aabb_intersector.cuh:17-21): 1 / d with |d| clamped away from 0 -- the operand is a
    // unit vector's component, FLT_EPSILON <= |x| <= 1, where rcp_exact_normal IS the IEEE quotient (rt_device.h) -- and
    // the scaled origin (-o) * (1 / d)
#ifndef RT_VERIFY_CLAMP
    // (a component below FLT_EPSILON in magnitude -- where the reference clamps, and where a -0.0 would sit -- is left to the
    // literal forms of the rare path: one min3 and one compare instead of three clamps and three sign tests)
    const bool tiny = fminf(fabsf(d.x), fminf(fabsf(d.y), fabsf(d.z))) < kFltEps;
    const V3 inv = mk(rcp_exact_normal(d.x), rcp_exact_normal(d.y), rcp_exact_normal(d.z));
#else
    const bool tiny = neg_zero3(d);
    const V3 inv = mk(rcp_exact_normal((fabsf(d.x) < kFltEps) ? copysignf(kFltEps, d.x) : d.x),
                      rcp_exact_normal((fabsf(d.y) < kFltEps) ? copysignf(kFltEps, d.y) : d.y),
                      rcp_exact_normal((fabsf(d.z) < kFltEps) ? copysignf(kFltEps, d.z) : d.z));
#endif
    const V3 so = mul(neg(o), inv);
    // the triangle's own box (triangle.cuh:9-10,22-37)
    const V3 p1 = sub(tr.p0, tr.e1), p2 = add(tr.p0, tr.e2);
    const float lox = fminf(tr.p0.x, fminf(p1.x, p2.x)), hix = fmaxf(tr.p0.x, fmaxf(p1.x, p2.x));
    const float loy = fminf(tr.p0.y, fminf(p1.y, p2.y)), hiy = fmaxf(tr.p0.y, fmaxf(p1.y, p2.y));
    const float loz = fminf(tr.p0.z, fminf(p1.z, p2.z)), hiz = fmaxf(tr.p0.z, fmaxf(p1.z, p2.z));
    // aabb_intersector.cuh:24-35: inv * bound + scaled_origin, a multiplication and an addition rounded one by one.  The
    // reference picks the near / far bound by the octant; with the octant consistent with the sign of 1 / d (no -0.0
    // component) the near bound's term is the smaller of the two (monotone rounding again), so min / max pick the same
    // values without the three compares and six selects
    const float tlx = inv.x * lox + so.x, thx = inv.x * hix + so.x;
    const float tly = inv.y * loy + so.y, thy = inv.y * hiy + so.y;
    const float tlz = inv.z * loz + so.z, thz = inv.z * hiz + so.z;
    const float entry = fmaxf(fminf(tlx, thx), fmaxf(fminf(tly, thy), fminf(tlz, thz)));
    const float exit = fminf(fmaxf(tlx, thx), fminf(fmaxf(tly, thy), fmaxf(tlz, thz)));
    bool vis = entry <= exit && !tiny;
    if (!vis) {  // rare (~1 hit in 10^7): the literal forms from here on
        vis = sc.ref_root_leaf != 0;  // bvh.cuh:252 / :307: a root that is a leaf is intersected without any box test
        if (!vis) {
            const RefSlab s = ref_slab(o, d);
            float e;
            // (the own box once more, literally: what the shortcut above could not decide -- a clamped or -0.0 component)
            vis = !neg_zero3(d) && ref_box(s, make_float4(lox, hix, loy, hiy), make_float4(loz, hiz, 0.f, 0.f), e);
        }
        if (!vis) {
            atomicAdd(&vstat[V_OWN_FAIL], 1ull);
            const RefSlab s = ref_slab(o, d);
            float e;
            int node = sc.ref_leaf_of[(unsigned)k];
            vis = ref_box(s, sc.ref_nodes[2 * node], sc.ref_nodes[2 * node + 1], e);
            if (vis && neg_zero3(d)) {  // no nesting argument for this ray: every ancestor below the root (the root's box is never tested)
#pragma nounroll
                for (int guard = 0; guard < 64 && vis; guard++) {
                    node = sc.ref_parent[(unsigned)node];
                    if (node <= 0) break;
                    vis = ref_box(s, sc.ref_nodes[2 * node], sc.ref_nodes[2 * node + 1], e);
                }
            }
            if (!vis) atomicAdd(&vstat[V_LOST], 1ull);
        }
        __asm__ volatile("" ::: "memory");
    }
    return vis;
}

enum { MODE_POOL = 0, MODE_TEST_CLOSEST = 2, MODE_TEST_ANY = 3 };

struct TraceParams {
    int total;             // number of slots (MODE_POOL) or test rays
    int debug_no_deposit;  // perf experiments only: skip the framebuffer atomics
    int fb_fixed;          // framebuffer holds 64-bit fixed-point sums (see deposit())
    float *fb;             // MODE_POOL: raw-sum framebuffer
    DWaveRow *rows;        // MODE_POOL: counter rows
    unsigned long long *prof;  // RT_TRACE_PROFILE builds only
    // lockstep rounds: nothing to trace in a round that shaded nothing (see k_advance); null / 0 otherwise
    const unsigned int *lock_shades;
    int lock_round;
    // test modes
    const float *o3, *d3, *tmax;
    const int *order, *excluded;
    int *out_i;
    float *out_t, *out_u, *out_v;
    unsigned long long *vstat;  // VERIFY builds: DCounters::vstat
};

// MODE_POOL traces BOTH ray kinds of a round in one launch: the path ray of every live slot
// (closest hit, ch()) and the shadow ray of every slot that spawned one (any hit, ah()).  A lane
// carries its kind with its ray, so closest-hit and any-hit rays share waves; the two kinds differ
// only in what a triangle hit does and in how the finished ray is finalised.
// LDS layout (dynamic): [stack: (stack_cap + 1) x kBlock ints (push_if)][pending: kBlock ints]
// MINW: minimum waves per SIMD the register budget is sized for (8 = 64 VGPRs: the renderer's build; the split probe
// also times the builds with 80 / 96 / 128 VGPRs).
// LITERAL (RT_FLAG_REFERENCE_WALK): a lane traverses its whole ray with reference_walk -- the scheduling around it
// (chunks, refill, finalisation) is unchanged, WIDE is not looked at.
// VERIFY (the default; off with RT_FLAG_WATERTIGHT): the product's walk with the reference's decisions -- see ref_visible.
template <int MODE, bool WIDE, int MINW = 8, bool LITERAL = false, bool VERIFY = false>
__global__ void __launch_bounds__(kBlock, MINW) k_trace(DScene sc, DPools p, TraceParams tp, int stack_cap, int *overflow) {
    if (MODE == MODE_POOL && tp.lock_shades != nullptr && tp.lock_round >= 1 && tp.lock_shades[tp.lock_round] == 0u) return;
    extern __shared__ int s_lds[];
    int *stack = s_lds + threadIdx.x;
    int *over = overflow + (blockIdx.x * kBlock + threadIdx.x);
    volatile int *pend = s_lds + (stack_cap + 1) * kBlock + (threadIdx.x & ~63);  // this wave's 64 entries
    const int total = tp.total;
    const int n_chunks = (total + 63) >> 6;                            // chunks per ray kind
    const int all_chunks = MODE == MODE_POOL ? 2 * n_chunks : n_chunks;  // [closest chunks][any chunks]
    const unsigned lane = lane_id();
    constexpr int kAnyBit = 1 << 30;  // ray id = slot | kAnyBit for shadow rays

    // wave-uniform work bookkeeping.  Candidates come in chunks of 64 consecutive slots, dealt
    // round-robin over the waves of the grid (chunk = wave id + k * waves): no shared head counter
    // -- a same-address atomic costs ~5 ns on this chip and 16k of them per launch formed a convoy.
    // The valid candidates of a chunk (live slots / slots that spawned a shadow ray this round) are
    // compacted into `pend` with ballot + mbcnt and handed to idle lanes from there, so the ray
    // queues of the reference (flag arrays + cub::DeviceSelect, render.cuh:431-443) exist only as
    // 64 ints of LDS per wave.
    int pend_lo = 0, pend_hi = 0;
    int next_chunk = (int)wave_index();
    const int grid_waves = (int)(gridDim.x * (kBlock / 64));
    bool exhausted = false;
    // per-lane ray state.  `tri` is the best hit so far (closest) or the excluded triangle (any);
    // `hu` doubles as the occluded flag of an any-hit ray.
    int id = -1, cur = kEntryDone, sp = 0, tri = -1;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 0), inv = mk(0, 0, 0);
    float tmax = 0.f, hu = 0.f, hv = 0.f;
    unsigned long long deposits = 0;
#ifdef RT_TRACE_PROFILE
    unsigned long long pf_outer = 0, pf_refill = 0, pf_inner_it = 0, pf_inner_lanes = 0, pf_leaf_it = 0, pf_leaf_lanes = 0,
                       pf_tri_it = 0, pf_tri_lanes = 0, pf_act_at_top = 0, pf_fin_lanes = 0, pf_new_lanes = 0;
#endif

    while (true) {
        unsigned long long act = wave_ballot(id >= 0 && cur != kEntryDone);
#ifdef RT_TRACE_PROFILE
        pf_outer++;
        pf_act_at_top += __popcll(act);
#endif
        if (__popcll(act) <= kRefillAt) {
            // ---- finalise finished lanes
            const bool fin = id >= 0 && cur == kEntryDone;
#ifdef RT_TRACE_PROFILE
            pf_refill++;
            pf_fin_lanes += wave_count((fin));
#endif
            const bool is_any = MODE == MODE_POOL ? (id & kAnyBit) != 0 : MODE == MODE_TEST_ANY;
            if (VERIFY && !LITERAL && fin && !is_any && tri >= 0) {
                // the closest hit stands if the reference's walk can see its triangle and nothing tied with it at the final
                // distance (the sign of hv: see the leaf phase); otherwise (~2 rays in 10^7) the ray is re-traced literally
                bool bad = (__float_as_uint(hv) >> 31) != 0u;
                if (bad) {
                    atomicAdd(&tp.vstat[V_TIE], 1ull);
                } else {
                    const Tri tr = load_tri(sc.tris, tri);
                    bad = !ref_visible(sc, o, d, tr, tri, tp.vstat);
                }
                if (bad) {
                    atomicAdd(&tp.vstat[V_LITERAL], 1ull);
                    tmax = MODE == MODE_POOL ? kFltMax : tp.tmax[id & (kAnyBit - 1)];
                    tri = -1;
                    hu = hv = 0.f;
                    reference_walk<false>(sc, o, d, tmax, tri, hu, hv, stack, over, stack_cap);
                }
            }
            if (MODE == MODE_POOL) deposits += wave_count((fin && is_any && hu == 0.f));
            if (fin) {
                const int slot = id & (kAnyBit - 1);
                if (MODE == MODE_POOL) {
                    if (!is_any) {
                        // hit record in the form mat() consumes (render.cuh:152-153, 311-316)
                        int info = -1;
                        if (tri >= 0) {
                            Tri tr = load_tri(sc.tris, tri);
                            int2 ml = sc.tri_info[(unsigned)tri];
                            V3 hp = tri_point(tr, hu, hv);
                            V3 hn = neg(unit(tr.n));
                            p.hpx(slot) = hp.x;
                            p.hpy(slot) = hp.y;
                            p.hpz(slot) = hp.z;
                            p.hnx(slot) = hn.x;
                            p.hny(slot) = hn.y;
                            p.hnz(slot) = hn.z;
                            info = (ml.x & 0xffff) | ((ml.y + 1) << 16);
                        }
                        p.hit_info(slot) = info;
                    } else if (hu == 0.f && !tp.debug_no_deposit) {  // unoccluded: render.cuh:291-293
                        int pixel = p.pixel(slot);
                        deposit(tp.fb, tp.fb_fixed, pixel, p.slr(slot), p.slg(slot), p.slb(slot));
                    }
                } else if (MODE == MODE_TEST_CLOSEST) {
                    tp.out_i[slot] = tri >= 0 ? tp.order[tri] : -1;
                    tp.out_t[slot] = tri >= 0 ? tmax : 0.f;
                    tp.out_u[slot] = hu;
                    tp.out_v[slot] = hv;
                } else {
                    tp.out_i[slot] = hu != 0.f ? 1 : 0;
                }
                id = -1;
            }
            // ---- refill idle lanes (up to three chunks per refill: shadow rays are sparse)
            for (int tries = 0; tries < 3; tries++) {
                unsigned long long idle = wave_ballot(id < 0);
                int n_idle = __popcll(idle);
                if (n_idle == 0) break;
                if (pend_lo == pend_hi && !exhausted) {
                    int chunk = next_chunk;
                    next_chunk += grid_waves;
                    if (chunk >= all_chunks) {
                        exhausted = true;
                    } else {
                        const bool any_chunk = MODE == MODE_POOL && chunk >= n_chunks;
                        int cand = (any_chunk ? chunk - n_chunks : chunk) * 64 + (int)lane;
                        bool valid = cand < total;
                        if (MODE == MODE_POOL && valid)
                            valid = any_chunk ? p.stmax(cand) >= 0.f : (p.bounces(cand) != kDone && p.bounces(cand) != kParked);
                        unsigned long long vm = wave_ballot(valid);
                        if (valid) pend[prefix_popc(vm)] = any_chunk ? (cand | kAnyBit) : cand;
                        pend_lo = 0;
                        pend_hi = __popcll(vm);
                    }
                }
                int avail = pend_hi - pend_lo;
                if (avail > 0) {
                    int r = (int)prefix_popc(idle);
                    if (id < 0 && r < avail) {
                        int my = pend[pend_lo + r];
                        int slot = my & (kAnyBit - 1);
                        if (MODE == MODE_POOL) {
                            if (my & kAnyBit) {
                                o = mk(p.sox(slot), p.soy(slot), p.soz(slot));
                                d = mk(p.sdx(slot), p.sdy(slot), p.sdz(slot));
                                tmax = p.stmax(slot);
                                tri = p.starget(slot);
                            } else {
                                o = mk(p.ox(slot), p.oy(slot), p.oz(slot));
                                d = mk(p.dx(slot), p.dy(slot), p.dz(slot));
                                tmax = kFltMax;
                                tri = -1;
                            }
                        } else {
                            o = mk(tp.o3[3 * slot], tp.o3[3 * slot + 1], tp.o3[3 * slot + 2]);
                            d = mk(tp.d3[3 * slot], tp.d3[3 * slot + 1], tp.d3[3 * slot + 2]);
                            tmax = tp.tmax[slot];
                            tri = MODE == MODE_TEST_ANY ? tp.excluded[slot] : -1;
                        }
                        id = my;
                        inv = inv_dir(d);
                        cur = 0;  // root pair
                        sp = 0;
                        hu = 0.f;
                    }
                    pend_lo += min(avail, n_idle);
                } else if (exhausted) {
                    break;
                }
            }
            act = wave_ballot(id >= 0 && cur != kEntryDone);
            if (act == 0) {
                if (exhausted && pend_lo == pend_hi) break;  // nothing in flight, nothing pending, no chunks left
                continue;
            }
        }
        if (LITERAL) {
            if (cur >= 0) {
                const bool is_any = MODE == MODE_POOL ? (id & kAnyBit) != 0 : MODE == MODE_TEST_ANY;
                if (is_any) reference_walk<true>(sc, o, d, tmax, tri, hu, hv, stack, over, stack_cap);
                else reference_walk<false>(sc, o, d, tmax, tri, hu, hv, stack, over, stack_cap);
                cur = kEntryDone;
            }
            continue;
        }
        // ---- inner phase: step through node records until no lane holds an inner entry
        while (wave_ballot(cur >= 0) != 0) {
#ifdef RT_TRACE_PROFILE
            pf_inner_it++;
            pf_inner_lanes += wave_count((cur >= 0));
#endif
            if (cur >= 0) inner_step<WIDE>(sc, o, inv, tmax, cur, sp, stack, over, stack_cap);
        }
        // ---- leaf phase: every lane that holds a leaf tests its triangles (triangle.cuh:39-58)
#ifdef RT_TRACE_PROFILE
        {
            unsigned long long lm = wave_ballot(cur != kEntryDone && cur < 0);
            if (lm) {
                pf_leaf_it++;
                pf_leaf_lanes += __popcll(lm);
                int cnt_l = (cur != kEntryDone && cur < 0) ? ((~cur) & 7) : 0;
                int mx = cnt_l, sm = cnt_l;
                for (int off = 32; off > 0; off >>= 1) { mx = max(mx, __shfl_xor(mx, off)); sm += __shfl_xor(sm, off); }
                pf_tri_it += mx;
                pf_tri_lanes += sm;
            }
        }
#endif
        if (cur != kEntryDone && cur < 0) {
            const bool is_any = MODE == MODE_POOL ? (id & kAnyBit) != 0 : MODE == MODE_TEST_ANY;
            int ref = ~cur;
            int first = ref >> 3, count = ref & 7;
            bool stop = false;
            for (int k = first; k < first + count; k++) {
                Tri tr = load_tri(sc.tris, k);
                float t, u, v;
                if (tri_intersect(tr, o, d, tmax, t, u, v)) {
                    if (is_any) {
                        // bvh.cuh:243: first accepted hit that is not the excluded triangle (VERIFY: and that the reference's
                        // walk can see at all)
                        if (k != tri && (!VERIFY || ref_visible(sc, o, d, tr, k, tp.vstat))) {
                            hu = 1.f;    // occluded
                            stop = true;
                            break;
                        }
                    } else {
                        const bool tie = t == tmax && tri >= 0;
                        if (closest_hit_wins(sc, t, tmax, k, tri)) {  // bvh.cuh:227-231 (t <= tmax)
                            tmax = t;
                            hu = u;
                            hv = v;
                            tri = k;
                        }
                        // VERIFY: an exact tie is the reference's tree order to decide (triangle.cuh:49): marked in the
                        // sign of hv (v >= 0 for an accepted hit; a closer hit later overwrites the mark with its own v)
                        if (VERIFY && tie) hv = __uint_as_float(__float_as_uint(hv) | 0x80000000u);
                    }
                }
            }
            if (stop) {
                cur = kEntryDone;
            } else if (sp > 0) {
                cur = stack_pop(stack, over, sp, stack_cap);
            } else {
                cur = kEntryDone;
            }
        }
    }
    if (MODE == MODE_POOL) {
        if (deposits != 0 && lane == 0) atomicAdd(&tp.rows[wave_index()].c[C_SHADOW_ADD], deposits);
    }
#ifdef RT_TRACE_PROFILE
    if (MODE == MODE_POOL && lane == 0 && tp.prof) {
        atomicAdd(&tp.prof[0], pf_outer); atomicAdd(&tp.prof[1], pf_refill); atomicAdd(&tp.prof[2], pf_inner_it);
        atomicAdd(&tp.prof[3], pf_inner_lanes); atomicAdd(&tp.prof[4], pf_leaf_it); atomicAdd(&tp.prof[5], pf_leaf_lanes);
        atomicAdd(&tp.prof[6], pf_tri_it); atomicAdd(&tp.prof[7], pf_tri_lanes); atomicAdd(&tp.prof[8], pf_act_at_top);
        atomicAdd(&tp.prof[9], pf_fin_lanes); atomicAdd(&tp.prof[10], 1ull);
    }
#endif
}

// ============================================================================ k_paths
// The whole asynchronous part of a frame in ONE launch.  A lane owns one path slot for the entire
// render and keeps its state in registers; the reference's stage kernels become PHASES of the lane:
//     ADV      init() + mat()                (advance_core; + gen() on small shards)
//     GEN      gen()                         (gen_core)
//     ANY      ah():  the slot's shadow ray  (any hit, deposit if unoccluded)
//     CLOSEST  ch():  the slot's path ray    (closest hit -> hit record for the next ADV)
// Because a slot never depends on another slot (see the file header) there is no barrier of any
// kind between rounds: a wave simply keeps all 64 of its slots moving until each has run out of
// camera rays (or parks for the lockstep final generation).  That removes what dominated the
// per-round design -- every k_trace launch ended in a drain where a wave waited for its longest
// ray with ~10 of 64 lanes active, 1 700 times per frame -- together with the per-round state
// traffic (rays, hit records and shadow rays never leave registers) and 3 400 kernel launches.
// Divergence between phases is handled by wave-level scheduling: each iteration the wave issues ONE
// block for all its lanes -- the expensive ADV block when at least `adv_batch` lanes wait for it (or
// nothing else can run), the short GEN block (gen() alone, for paths that certainly ended) when
// `gen_batch` lanes wait for it, otherwise the more popular of a node block (up to 8 node steps) and a
// triangle block (up to 2 tests).
enum { PH_ADV = 0, PH_ANY = 1, PH_CLOSEST = 2, PH_IDLE = 3, PH_GEN = 4 };
#ifndef RT_TRI_PER_STEP
#define RT_TRI_PER_STEP 2
#endif
constexpr int kCidChunk = 512;  // camera-ray ids a wave draws at a time in the per-sample RNG mode
constexpr int kTriPerStep = RT_TRI_PER_STEP;  // triangle tests a lane makes per scheduled triangle block
#ifndef RT_NODE_PER_STEP
#define RT_NODE_PER_STEP 8
#endif
constexpr int kNodePerStep = RT_NODE_PER_STEP;  // node steps a lane makes per scheduled node block (2-wide records)
#ifndef RT_NODE_PER_STEP_WIDE
#define RT_NODE_PER_STEP_WIDE 2
#endif
constexpr int kNodePerStepWide = RT_NODE_PER_STEP_WIDE;  // ... with 4-wide nodes (measured: 3 649 / 3 620 / 3 539 / 3 391 Msamples/s at 2 / 3 / 4 / 5)
// 4-wide node blocks are adaptive: kNodePerStepWide steps for every lane that has one to make, then -- if at least
// kNodeCont lanes of the wave still do -- up to kNodeExtra more (a wave-uniform branch).  Measured on the four BASELINE
// scenes against fixed 2 / 3 / 4 / 5 steps: fixed 4 is 8 % faster on the sixteen-light scene (its shadow rays make
// 4.8 node steps against 2.2 in C2) and 5 % slower on the matte scene; 2 + 2 at >= 36 lanes is at least as fast as
// fixed 2 on all four.  A `do ... while (enough lanes)` loop that is not unrolled loses 1 %.
#ifndef RT_NODE_CONT
#define RT_NODE_CONT 36
#endif
#ifndef RT_NODE_EXTRA
#define RT_NODE_EXTRA 2
#endif
constexpr int kNodeCont = RT_NODE_CONT, kNodeExtra = RT_NODE_EXTRA;
#ifndef RT_SPECULATE
#define RT_SPECULATE 1
#endif

constexpr bool kSpeculate = RT_SPECULATE != 0;  // k_paths: postpone a leaf reached inside a node block (see `pend` there)

// Register diet: across loop iterations a lane carries only ONE ray (o, d, 1/d, tmax) and the
// traversal cursor (cur, sp, tri, hu, hv).  The slot's persistent state (bounces, pixel, gen, RNG,
// beta = 12 dwords) lives in the lane's LDS column and is only in registers inside the ADV block;
// while a shadow ray is traced, the slot's path ray and the radiance to deposit wait in 9 more
// dwords of LDS; the hit record is rebuilt from (tri, hu, hv) inside the ADV block.
// LDS layout (dynamic): [stack: (stack_cap + 1) x kBlock (push_if)][parked ray: 9 x kBlock][slot state: 12 + 1 x kBlock][sample sum: 3 x kBlock][tables]
// LITERAL (RT_FLAG_REFERENCE_WALK): the node block is a lane's WHOLE ray through reference_walk -- the reference's tree,
// box test, order and tie rule; no triangle blocks, no speculation.  Everything around it (phases, ADV / GEN blocks, the
// sample accumulator) is the same code.  Opt-in and never timed.
// VERIFY (the default build; off with RT_FLAG_WATERTIGHT): the reference's decisions on this kernel's own walk (see
// ref_visible): a shadow ray's accepted hit counts only if the reference's walk can see its triangle (triangle block); a
// path ray's closest hit is checked once, at the top of the ADV block that shades it -- visible, and no exact tie at the
// final distance -- and the ~2 rays in 10^7 that fail are re-traced there by reference_walk.
template <bool LDS_TABLES, bool WIDE, bool MAJORITY, int MIN_WAVES, bool DRAW_CIDS = false, bool LITERAL = false, bool VERIFY = false>
__global__ void __launch_bounds__(kBlock, MIN_WAVES)
k_paths(DScene sc, DPools p, Camera cam_arg, AdvanceParams ap_arg, float *__restrict__ fb, DWaveRow *__restrict__ rows,
        int stack_cap, int *overflow, int adv_batch, int debug_no_deposit, unsigned long long *prof, int top_n,
        int prio_period, int rot_wave, int rot_set, int gen_batch, int tri_follow, unsigned int *__restrict__ next_cid,
        int half_fill, unsigned long long *__restrict__ vstat) {
    // The GEN block exists where the chip is short of issue slots (4 waves per SIMD): there it takes a third of the
    // lanes out of the long ADV block (+3 %, and the ADV block no longer spills).  On small shards (2 waves per
    // SIMD) a slot-round is a latency chain and one more block in it costs 5 %: gen() stays inside ADV there.
    constexpr bool SPLIT_GEN = MIN_WAVES != 2;
    extern __shared__ int s_lds[];
    int *stack = s_lds + threadIdx.x;
    float *park = (float *)(s_lds + (stack_cap + 1) * kBlock) + threadIdx.x;  // element k at park[k * kBlock]
    int *over = overflow + (blockIdx.x * kBlock + threadIdx.x) % kOverStride;
    int *cold = s_lds + (stack_cap + 10) * kBlock + threadIdx.x;  // element k at cold[k * kBlock]
    float *acc = (float *)(s_lds + (stack_cap + 23) * kBlock) + threadIdx.x;  // sample accumulator (acc_add / acc_flush)
    float *s_tab = (float *)(s_lds + (stack_cap + 26) * kBlock);
    const float *tab = sc.tables;
    // small shards (MIN_WAVES == 2: at most 2 workgroups per CU, LDS to spare, latency-bound): the top of
    // the BVH is staged in LDS, so the first levels of every traversal do not leave the CU
    const float4 *s_top = (const float4 *)(s_tab + (LDS_TABLES ? ((sc.tab_dwords + 3) & ~3) : 0));  // (tables: what the scene needs)
    if (MIN_WAVES != 2) top_n = 0;
    for (int k = threadIdx.x; k < top_n * 4; k += kBlock) ((float4 *)s_top)[k] = sc.nodes[k];
    if (LDS_TABLES) {
        for (int k = threadIdx.x; k < sc.tab_dwords; k += kBlock) s_tab[k] = sc.tables[k];
        tab = s_tab;
    }
    // The camera and the frame parameters are only needed inside the GEN / ADV blocks: kept as kernel arguments
    // they occupy ~30 SGPRs for the whole loop and push other scalars out into VGPR lanes (v_readlane /
    // v_writelane are VALU work).  Staged in LDS they are read where they are used.
    struct Uniforms {
        Camera cam;
        AdvanceParams ap;
    };
    static_assert(sizeof(Uniforms) % 4 == 0, "dword copy");
    Uniforms *s_uni = (Uniforms *)(s_top + 4 * (size_t)top_n);
    {
        Uniforms u;
        u.cam = cam_arg;
        u.ap = ap_arg;
        const int *srcw = (const int *)&u;
        for (int k = threadIdx.x; k < (int)(sizeof(Uniforms) / 4); k += kBlock) ((int *)s_uni)[k] = srcw[k];
    }
    __syncthreads();
    const Camera &cam = s_uni->cam;
    const AdvanceParams &ap = s_uni->ap;
    // what the scheduling loop itself needs stays scalar
    const int ap_n = ap_arg.n, ap_max_bounces = ap_arg.max_bounces, ap_fb_fixed = ap_arg.fb_fixed;
    // RT_FLAG_RNG_PER_SAMPLE: camera rays are not tied to slots, so the wave DRAWS them -- kCidChunk ids at a time from the
    // frame's counter (one global atomic per chunk), handed to its lanes as they ask for one -- and no lane is left with more
    // work than the others at the end of the frame (the static deal of slots costs 6 % there, 23 % on a 1/8 frame)
    // (a build of its own -- DRAW_CIDS -- so that the reference-mode kernel carries none of it)
    constexpr bool draw_cids = DRAW_CIDS && SPLIT_GEN;
    int cid_next = 0, cid_end = 0;
    // A lane works through the slots i, i + G, i + 2G, ... (G = lanes of the grid), each for the whole
    // frame, one after the other: with G dividing the slot count every lane gets the same number of
    // slots, so all lanes -- and all workgroups, which are all resident -- finish together.
    const int lanes_in_grid = (int)(gridDim.x * blockDim.x);
    // Which slots.  Slot s works through the pixels (s + g W) / spp, g = 0, 1, ...: a lattice of a few image
    // columns (every 128th at 1920 x 1080 x 256 spp), the same lattice for slots s and s + 64 * lattice_blocks.
    // A wave always owns 64 CONSECUTIVE slots (samples of one pixel: coherent primary rays; splitting waves
    // into quarters was measured and loses more than it balances), but with plain striding the four waves of a
    // workgroup, the four workgroups of a CU and all successive slot sets of a lane fall on ONE lattice, and
    // whole CUs differ by +-5 % in work (bunny or no bunny in their columns) -- which the slowest one turns
    // into frame time.  So the j-th wave of a workgroup is shifted by j quarter periods and the k-th slot set
    // by k * 5/16 of a period.  A bijection between (set, 64-slot block) and (set, wave).  +6 % at 1 GPU.
    auto slot_of = [&](int set) {  // (everything recomputed here: nothing of this lives across the main loop)
        const unsigned lane_in_grid = blockIdx.x * blockDim.x + threadIdx.x;
        const unsigned wave_in_grid = lane_in_grid >> 6, lane_in_wave = lane_in_grid & 63u;
        // (the grid is a power of two: W is, shard counts divide it, and the host halves from there)
        const unsigned b = (wave_in_grid + (wave_in_grid & 3u) * (unsigned)rot_wave + (unsigned)set * (unsigned)rot_set) &
                           (((unsigned)lanes_in_grid >> 6) - 1u);
        // half_fill (small shards, experiment): only lanes 0..31 of a wave own slots -- twice the waves, each with half
        // the chains: a wave's block stream gets shorter where issue slots are to spare
        if (half_fill) return lane_in_wave < 32u ? set * (lanes_in_grid >> 1) + (int)(b * 32u + lane_in_wave) : 0x7fffffff;
        return set * lanes_in_grid + (int)(b * 64u + lane_in_wave);
    };
    int slot_set = 0;
    int i = slot_of(0);
    // persistent slot state
    int bounces = kDone, pixel = 0, gen = 0;
    Rng rs{0, 0, 0, 0, 0, 0};
    V3 beta = mk(0, 0, 0);
    auto load_slot = [&](int k) {
        bounces = p.bounces(k);
        pixel = p.pixel(k);
        gen = p.gen(k);
        rs = Rng{p.rd(k), p.r0(k), p.r1(k), p.r2(k), p.r3(k), p.r4(k)};
        beta = mk(p.br(k), p.bg(k), p.bb(k));
    };
    // hand a finished slot back to the pools: the lockstep rounds of the final generation continue from there
    auto store_slot = [&](int k) {
        p.bounces(k) = bounces;
        p.pixel(k) = pixel;
        p.gen(k) = gen;
        p.hit_info(k) = -1;
        p.stmax(k) = -1.f;
        p.br(k) = beta.x;
        p.bg(k) = beta.y;
        p.bb(k) = beta.z;
        p.rd(k) = rs.d;
        p.r0(k) = rs.v0;
        p.r1(k) = rs.v1;
        p.r2(k) = rs.v2;
        p.r3(k) = rs.v3;
        p.r4(k) = rs.v4;
    };
    auto cold_save = [&]() {
        cold[0 * kBlock] = bounces;
        cold[1 * kBlock] = pixel;
        cold[2 * kBlock] = gen;
        cold[3 * kBlock] = (int)rs.d;
        cold[4 * kBlock] = (int)rs.v0;
        cold[5 * kBlock] = (int)rs.v1;
        cold[6 * kBlock] = (int)rs.v2;
        cold[7 * kBlock] = (int)rs.v3;
        cold[8 * kBlock] = (int)rs.v4;
        cold[9 * kBlock] = __float_as_int(beta.x);
        cold[10 * kBlock] = __float_as_int(beta.y);
        cold[11 * kBlock] = __float_as_int(beta.z);
    };
    auto cold_load = [&]() {
        bounces = cold[0 * kBlock];
        pixel = cold[1 * kBlock];
        gen = cold[2 * kBlock];
        rs = Rng{(uint32_t)cold[3 * kBlock], (uint32_t)cold[4 * kBlock], (uint32_t)cold[5 * kBlock],
                 (uint32_t)cold[6 * kBlock], (uint32_t)cold[7 * kBlock], (uint32_t)cold[8 * kBlock]};
        beta = mk(__int_as_float(cold[9 * kBlock]), __int_as_float(cold[10 * kBlock]), __int_as_float(cold[11 * kBlock]));
    };
    int phase = PH_IDLE;
    int tri = -1;
    // the ray being traced, and the traversal cursor (`tri`: best hit so far / excluded triangle;
    // `hu` doubles as the occluded flag of a shadow ray, exactly as in k_trace).  Between the end of
    // a closest-hit trace and the ADV block, (tri, hu, hv, d) ARE the hit record.
    V3 o = mk(0, 0, 0), d = mk(0, 0, 0), inv = mk(0, 0, 0);
    float tmax = 0.f, hu = 0.f, hv = 0.f;
    int cur = kEntryDone, sp = 0;
    // hand a slot that has no camera ray left (or is parked for the lockstep final generation) back to the pools and take
    // the lane's next one; `first_phase`: what an untouched slot does first (its bounces = INT_MAX makes that a gen())
    auto next_slot = [&](int first_phase) {
        cold_load();
        store_slot(i);
        phase = PH_IDLE;
        tri = -1;
        slot_set++;
        i = slot_of(slot_set);
        if (i < ap_n) {
            load_slot(i);
            if (bounces != kDone && bounces != kParked) {
                phase = first_phase;
                cold_save();
                cold[12 * kBlock] = -1;
            } else {
                i = ap_n;  // (cannot happen: untouched slots start alive)
            }
        }
    };
    // Speculative traversal (kSpeculate): a lane that reaches a leaf inside a node block does not stop there -- it sets
    // the leaf aside in `pend` and goes on with the next stack entry, so the up-to-8 node steps of a block are used by
    // most lanes to the end (without it half of them idle from the middle of the block on), and the triangle block
    // finds two leaves per lane.  The triangles of the postponed leaf are tested a little later, with whatever tmax the
    // ray has then: the closest hit is the minimum over the accepted hits whatever the order (ties: closest_hit_wins),
    // an occluder is an occluder whenever it is found; the price is a few node visits a fresher tmax would have culled.
    int pend = kEntryDone;
    acc[0 * kBlock] = acc[1 * kBlock] = acc[2 * kBlock] = 0.f;
    if (i < ap_n) {
        load_slot(i);
        phase = (bounces != kDone && bounces != kParked) ? (SPLIT_GEN ? PH_GEN : PH_ADV) : PH_IDLE;  // (untouched slots: bounces = INT_MAX)
        cold_save();
        cold[12 * kBlock] = -1;  // no previous pixel
    }
    // wave-uniform event counters
    unsigned long long n_gen = 0, n_shade = 0, n_traced = 0, n_shadow = 0, n_emit = 0, n_deposit = 0, n_rr = 0;
#ifdef RT_TRACE_PROFILE
    unsigned long long pf[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pf_gen_cycles = 0, pf_fin_cycles = 0, pf_fin_lanes = 0, pf_fin_iters = 0;
    const unsigned long long pf_t0 = __builtin_readcyclecounter();
#endif

    // The SIMD's issue arbiter prefers the OLDEST of its waves.  Left alone, the four waves of a SIMD (one from
    // each of the CU's four resident workgroups) finish in dispatch order at 0.63 / 0.73 / 0.81 / 0.92 of the
    // kernel's duration although they carry the same work, and the SIMD spends the last third of the frame with
    // three, two, one wave -- too few to hide anything.  Rotating the waves' priorities makes them progress
    // together (0.87 .. 0.92; +7 % throughput): every 2^prio_period scheduling decisions a wave takes the next of
    // the four levels, starting from its workgroup's residency rank.  (Steering the priority by measured
    // progress against the grid's average was tried and is worse than the plain rotation.)
    unsigned prio_tick = 0;
    const unsigned prio_rank = (4u * blockIdx.x) / gridDim.x;
    while (true) {
        RT_MARK("loop.head");
        if (prio_period && (prio_tick++ & ((1u << prio_period) - 1u)) == 0u) {
            unsigned lvl = ((prio_tick >> prio_period) + prio_rank) & 3u;
            // ties go to the older wave, which left the two younger waves of a SIMD 5 % behind the two older ones;
            // never dropping them to the lowest level evens that out (ranks finish at 0.95 .. 0.97 of the frame)
            lvl = max(lvl, prio_rank >> 1);
            switch (lvl) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        // ---- what each lane wants next: the ADV block, a node step, or triangle tests
        bool trav = phase == PH_ANY || phase == PH_CLOSEST;
        bool want_node = trav && cur >= 0;
        bool want_tri = trav && ((cur != kEntryDone && cur < 0) || (kSpeculate && pend != kEntryDone));
        int n_adv = wave_count((phase == PH_ADV));
        const int n_genw = wave_count((phase == PH_GEN));
        int n_node = wave_count((want_node));
        int n_tri = wave_count((want_tri));
        if (n_adv + n_genw + n_node + n_tri == 0) break;
        // After a GEN or ADV block the wave goes straight on to the traversal blocks of this scheduling round (its lanes have
        // just been given rays at the root): the lanes' wishes are taken again and the round trip through the loop head is
        // saved (+3.8 % on C2).
        auto retake = [&]() {
            trav = phase == PH_ANY || phase == PH_CLOSEST;
            want_node = trav && cur >= 0;
            want_tri = trav && ((cur != kEntryDone && cur < 0) || (kSpeculate && pend != kEntryDone));
            n_adv = wave_count((phase == PH_ADV));
            n_node = wave_count((want_node));
            n_tri = wave_count((want_tri));
        };
        // Every block is issued for the whole wave whatever the number of lanes that need it.  The ADV
        // block is ~15x longer than a node step or a triangle test, so it waits for `adv_batch` lanes
        // unless nothing else can run.  MAJORITY additionally runs only the more popular of the two
        // traversal blocks per iteration (one triangle per lane per iteration).
        bool run_adv = n_adv > 0 && (n_adv >= adv_batch || n_node + n_tri == 0);
        // (MAJORITY: the ADV lanes must also be at least half as many as the node and as the triangle lanes.  Requiring
        // a full majority measured 1 % slower.  Dropping the condition is as fast in logic, but when the kernel sat
        // exactly at 128 VGPRs that source shape tipped the register allocation into 21 spills: -6 %.  The kernel has
        // since come down to 115, but `make resource-usage` after any edit here all the same: "VGPRs Spill" of
        // k_paths<..., 4> must stay 0 (a CPU test checks it).)
        if (MAJORITY) run_adv = n_adv > 0 && ((n_adv >= adv_batch && 2 * n_adv >= n_node && 2 * n_adv >= n_tri) || n_node + n_tri == 0);
        // ---------------- GEN block: gen() (render.cuh:250-275) for the lanes whose path certainly ended -- it missed
        // or ran out of bounces (a third of all ADV work), or the ADV block found it Russian-roulette-killed to the
        // last bounce.  A tenth of the ADV block's length, so it runs for far fewer waiting lanes.
        if (SPLIT_GEN && !run_adv && n_genw > 0 && (n_genw >= gen_batch || n_node + n_tri == 0)) {
            RT_MARK("gen.begin");
#ifdef RT_TRACE_PROFILE
            pf[12]++; pf[15] += n_genw;
            const unsigned long long pf_tg = __builtin_readcyclecounter();
#endif
            // (what the block changes in the lane's loop-carried registers is applied by selects after the divergent part,
            // and the rare hand-back of a finished slot sits behind a wave-uniform branch: as assignments inside the
            // branches these cost the wave ~45 register moves per GEN block at the merges)
            AdvanceOut out;
            out.did_gen = out.new_ray = false;
            out.ray_o = o;
            out.ray_d = d;
            bool hand_back = false;
            long long my_cid = -1;
            if (draw_cids) {
                const unsigned long long m = wave_ballot(phase == PH_GEN);
                const int need = (int)__popcll(m), rank = (int)prefix_popc(m);
                int served = 0;
                while (served < need) {
                    if (cid_next == cid_end) {
                        unsigned base = 0;
                        if (lane_id() == 0) base = atomicAdd(next_cid, (unsigned)kCidChunk);
                        // (past the frame's end: "none left".  The clamp leaves room for cid_end = cid_next + kCidChunk below
                        // INT_MAX; ids that large are never rendered anyway: render_shard_impl refuses frames whose camera-ray
                        // ids come within 13 W of 2^31, and a wave overshoots the frame's end by at most one chunk)
                        static_assert(0x7ffff000u + (unsigned)kCidChunk < 0x7fffffffu, "cid_end must not overflow int");
                        cid_next = (int)min(__builtin_amdgcn_readfirstlane(base), 0x7ffff000u);
                        cid_end = cid_next + kCidChunk;
                    }
                    const int take = min(need - served, cid_end - cid_next);
                    if (phase == PH_GEN && rank >= served && rank < served + take) my_cid = cid_next + (rank - served);
                    cid_next += take;
                    served += take;
                }
            }
            if (phase == PH_GEN) {
                SlotState st;
                st.gen = cold[2 * kBlock];
                st.rs = Rng{(uint32_t)cold[3 * kBlock], (uint32_t)cold[4 * kBlock], (uint32_t)cold[5 * kBlock],
                            (uint32_t)cold[6 * kBlock], (uint32_t)cold[7 * kBlock], (uint32_t)cold[8 * kBlock]};
                st.bounces = 0;
                st.pixel = 0;
                st.beta = mk(0, 0, 0);
                int pxy = cold[12 * kBlock];
                acc_flush(acc, fb, ap_fb_fixed, cold[1 * kBlock]);  // the camera ray that ended: its sum -> its pixel
                gen_core<true>(cam, ap, ap.slot_lo + i, st, out, &pxy, my_cid);
                // the slot state gen() leaves: bounces (0, or the kDone / kParked sentinel), gen, the RNG; a new path also
                // has its pixel and beta = 1
                cold[0 * kBlock] = st.bounces;
                cold[2 * kBlock] = st.gen;
                cold[3 * kBlock] = (int)st.rs.d;
                cold[4 * kBlock] = (int)st.rs.v0;
                cold[5 * kBlock] = (int)st.rs.v1;
                cold[6 * kBlock] = (int)st.rs.v2;
                cold[7 * kBlock] = (int)st.rs.v3;
                cold[8 * kBlock] = (int)st.rs.v4;
                if (out.new_ray) {
                    cold[1 * kBlock] = st.pixel;
                    cold[9 * kBlock] = __float_as_int(st.beta.x);
                    cold[10 * kBlock] = __float_as_int(st.beta.y);
                    cold[11 * kBlock] = __float_as_int(st.beta.z);
                    cold[12 * kBlock] = pxy;
                } else {
                    hand_back = true;  // out of camera rays, or parked for the lockstep final generation
                }
            }
            const bool nr = out.new_ray;
            o = out.ray_o;
            d = out.ray_d;
            inv = inv_dir(d);  // (for every lane, as after the ADV block: the same value for the lanes that keep their ray)
            phase = nr ? (int)PH_CLOSEST : phase;
            tmax = nr ? kFltMax : tmax;
            tri = nr ? -1 : tri;
            cur = nr ? 0 : cur;
            sp = nr ? 0 : sp;
            if (wave_ballot(hand_back)) {  // rare (once per slot and frame): kept out of the merges above
                if (hand_back) {
                    if (draw_cids) phase = PH_IDLE;  // (the frame's counter has run out: nothing is tied to this lane's slot)
                    else next_slot(PH_GEN);
                }
            }
            n_gen += wave_count((out.did_gen));
            n_traced += wave_count((out.new_ray));
#ifdef RT_TRACE_PROFILE
            pf_gen_cycles += __builtin_readcyclecounter() - pf_tg;
#endif
            retake();
            RT_MARK("gen.end");
        }
        if (run_adv) {
#ifdef RT_TRACE_PROFILE
            pf[0]++; pf[1] += n_adv;
            const unsigned long long pf_ta = __builtin_readcyclecounter();
#endif
            // ---------------- ADV block
            RT_MARK("adv.head");
            AdvanceOut out;
            out.did_gen = out.did_shade = out.has_shadow = out.did_emit = out.new_ray = false;
            out.rr_draws = 0;
            if (phase == PH_ADV) {
                // the slot state is only live between cold_load() and cold_save() below
                cold_load();
                SlotState st;
                st.wo = d;
                st.hit_info = -1;
                st.isect_p = st.isect_n = mk(0, 0, 0);
                if (tri >= 0) {  // hit record in the form mat() consumes (render.cuh:152-153, 311-316)
                    Tri tr = load_tri(sc.tris, tri);
                    float4 sh = sc.tri_shade[(unsigned)tri];
                    RT_MARK("adv.verify");
#ifdef RT_DBG_NO_ADV_VERIFY  // (timing experiment only: the image is no longer the reference's)
                    if (false) {
#else
                    if (VERIFY) {
#endif
                        // (o, d) are still the path ray that ended on `tri`.  A set sign bit of hv: an exact tie at the final
                        // distance (triangle block) -- or a v of -0.0, which costs a needless, equally exact re-trace
                        bool bad = (__float_as_uint(hv) >> 31) != 0u;
                        if (bad) atomicAdd(&vstat[V_TIE], 1ull);
                        else bad = !ref_visible(sc, o, d, tr, tri, vstat);
                        if (bad) {
                            atomicAdd(&vstat[V_LITERAL], 1ull);
                            float tm = kFltMax;
                            tri = -1;
                            hu = hv = 0.f;
                            reference_walk<false>(sc, o, d, tm, tri, hu, hv, stack, over, stack_cap);
                            cold_load();  // (again: what was loaded above need not stay in registers across the walk)
                            if (tri >= 0) {
                                tr = load_tri(sc.tris, tri);
                                sh = sc.tri_shade[(unsigned)tri];
                            }
                        }
                    }
                    RT_MARK("adv.hit_record");
                    if (!VERIFY || tri >= 0) {
                        st.isect_p = tri_point(tr, hu, hv);
                        st.isect_n = mk(sh.x, sh.y, sh.z);
                        st.hit_info = __float_as_int(sh.w);
                    }
                }
                st.bounces = bounces;
                st.pixel = pixel;
                st.gen = gen;
                st.rs = rs;
                st.beta = beta;
                advance_core<SPLIT_GEN, true, true>(sc, tab, cam, ap, ap.slot_lo + i, st, out, fb, acc);
                RT_MARK("adv.tail");
                bounces = st.bounces;
                pixel = st.pixel;
                gen = st.gen;
                rs = st.rs;
                beta = st.beta;
                if (out.has_shadow) {
                    park[0 * kBlock] = out.ray_o.x;
                    park[1 * kBlock] = out.ray_o.y;
                    park[2 * kBlock] = out.ray_o.z;
                    park[3 * kBlock] = out.ray_d.x;
                    park[4 * kBlock] = out.ray_d.y;
                    park[5 * kBlock] = out.ray_d.z;
                    park[6 * kBlock] = out.s_L.x;
                    park[7 * kBlock] = out.s_L.y;
                    park[8 * kBlock] = out.s_L.z;
                    o = out.s_o;
                    d = out.s_d;
                    phase = PH_ANY;
                    tmax = out.s_tmax;
                    tri = out.s_target;
                    hu = 0.f;
                } else if (out.new_ray) {
                    o = out.ray_o;
                    d = out.ray_d;
                    phase = PH_CLOSEST;
                    tmax = kFltMax;
                    tri = -1;
                } else if (SPLIT_GEN) {
                    phase = PH_GEN;  // out.wants_gen: Russian roulette ended the path (its draws are in rs)
                    tri = -1;
                } else {
                    // this slot is out of camera rays (or parked for the lockstep final generation)
                    cold_save();
                    next_slot(PH_ADV);
                }
                if (phase == PH_ANY || phase == PH_CLOSEST) {
                    cur = 0;
                    sp = 0;
                }
                // (the same treatment as in the GEN block -- selects after the divergent part -- was measured here and
                // loses 1 %: the shading block's exits are three-way and the selects outnumber the moves they replace)
                if (phase != PH_IDLE) cold_save();
            }
            // 1 / d for EVERY lane, also those that only sat through the block: three v_rcp_f32, and 1 / d does not have
            // to stay in registers across the ~1 100 vector instructions of the block (three registers that decided
            // between a build with and without spills when the kernel sat at 128 VGPRs)
            inv = inv_dir(d);
            if (!SPLIT_GEN) n_gen += wave_count((out.did_gen));
            n_shade += wave_count((out.did_shade));
            n_traced += wave_count((out.new_ray));
            n_shadow += wave_count((out.has_shadow));
            n_emit += wave_count((out.did_emit));
            int rr = out.rr_draws;
            if (wave_ballot(rr != 0)) {
                for (int off = 32; off > 0; off >>= 1) rr += __shfl_xor(rr, off);
                n_rr += (unsigned long long)rr;
            }
#ifdef RT_TRACE_PROFILE
            pf[8] += __builtin_readcyclecounter() - pf_ta;
#endif
            retake();
            RT_MARK("adv.after");
        }
        const bool is_any = phase == PH_ANY;
        // ---------------- node steps for the lanes in `want`
        auto node_block = [&](bool want, int n_want) {
#ifdef RT_TRACE_PROFILE
            pf[2]++; pf[3] += n_want; pf[6] += n_adv;
            const unsigned long long pf_tn = __builtin_readcyclecounter();
#endif
            RT_MARK("node.begin");
            if (LITERAL) {
                if (want) {
                    if (is_any) reference_walk<true>(sc, o, d, tmax, tri, hu, hv, stack, over, stack_cap);
                    else reference_walk<false>(sc, o, d, tmax, tri, hu, hv, stack, over, stack_cap);
                    cur = kEntryDone;
                }
            } else if (want) {
                // a bounded while-while: up to kNodePerStep consecutive node steps (2 triangle tests in the
                // triangle block) per scheduling decision -- measured best at 8 / 2 (+22 % over 1 / 1; 4 / 2: +20 %)
                auto step_general = [&]() {
                    if (cur >= 0) {
                        inner_step<WIDE>(sc, o, inv, tmax, cur, sp, stack, over, stack_cap, s_top, top_n);
                    } else if (kSpeculate && cur != kEntryDone && pend == kEntryDone && sp > 0) {
                        pend = cur;  // a leaf: set it aside, go on with the next entry
                        cur = stack_pop(stack, over, sp, stack_cap);
                    }
                };
                // 4-wide nodes on the full pool: ONE wave vote per step decides between the step without any overflow
                // handling (all lanes of the block hold at most stack_cap - 3 entries: 95 % of the steps) and the general one
                auto step = [&]() {
#ifdef RT_NO_SHALLOW
                    if (true) {
#else
                    if (!WIDE || MIN_WAVES == 2) {
#endif
                        step_general();
                    } else if (wave_ballot(sp > stack_cap - 3) == 0ull) {
                        if (cur >= 0) {
                            inner_step<WIDE, true>(sc, o, inv, tmax, cur, sp, stack, over, stack_cap);
                        } else if (kSpeculate && cur != kEntryDone && pend == kEntryDone && sp > 0) {
                            pend = cur;
                            sp--;
                            cur = stack[sp * kBlock];
                        }
                    } else {
                        step_general();
                    }
                };
                if (kNodeCont == 0 || !WIDE) {
#pragma unroll
                    for (int rep = 0; rep < (WIDE ? kNodePerStepWide : kNodePerStep); rep++) step();
                } else {
                    // adaptive: the fixed steps, then kNodeExtra more if enough lanes of the wave still have one to make
#pragma unroll
                    for (int rep = 0; rep < kNodePerStepWide; rep++) step();
                    if (wave_count(cur >= 0) >= kNodeCont) {
#pragma unroll
                        for (int rep = 0; rep < kNodeExtra; rep++) step();
                    }
                }
            }
#ifdef RT_TRACE_PROFILE
            pf[9] += __builtin_readcyclecounter() - pf_tn;
#endif
            RT_MARK("node.end");
        };
        // ---------------- triangle tests (triangle.cuh:39-58) for the lanes in `want`: the leaf reference is the cursor
        auto tri_block = [&](bool want, int n_want) {
#ifdef RT_TRACE_PROFILE
            pf[4]++; pf[5] += n_want; pf[7] += n_adv;
            const unsigned long long pf_tt = __builtin_readcyclecounter();
#endif
            RT_MARK("tri.begin");
            if (want) {
                // kTriPerStep tests per lane, and ALL their triangle records are fetched before the first test: which
                // triangles come next does not depend on the outcome of a test (only whether they are still wanted does: an
                // occluded shadow ray is finished), so the block waits for one memory round trip instead of one per test.
                // The lane's walk through its leaves is made up front on copies of (pend, cur):
                //   the postponed leaf first, then the leaf under the cursor; when the cursor's leaf is used up, the next
                //   stack entry -- popped on the spot: a ray that turns out occluded has no use for its stack any more.
                // Straight-line bookkeeping: every outcome is a select, not a branch (the merges of the branchy version cost
                // the wave ~30 register moves per test); the only branches left are the rare ones (a tie between two hits;
                // a pop from the overflow column).
                int pd = pend, cu = cur;
                int ks[kTriPerStep] = {};  // (0 = a valid triangle address for lanes that have nothing to fetch)
                bool act[kTriPerStep];
                Tri tr[kTriPerStep];
#pragma unroll
                for (int j = 0; j < kTriPerStep; j++) {
                    const bool fp = kSpeculate && pd != kEntryDone;
                    const bool leaf = cu != kEntryDone && cu < 0;
                    act[j] = fp || leaf;
                    const int enc = fp ? pd : cu;  // ~((first << 3) | count)
                    ks[j] = act[j] ? (~enc) >> 3 : ks[0];  // (an address that is valid in any case)
                    const bool more = ((~enc) & 7) > 1;
                    const int rest = more ? enc - 7 : kEntryDone;  // one triangle further: first + 1, count - 1
                    int popped = kEntryDone;
                    // (the same wave vote as in the node step -- no lane has entries in the overflow part -- measured here: +1 %
                    // SLOWER, it undoes the node step's gain: profiles/r05_experiments.md)
                    if (act[j] && !fp && !more && sp > 0) popped = stack_pop(stack, over, sp, stack_cap);
                    pd = (act[j] && fp) ? rest : pd;
                    cu = (act[j] && !fp) ? (more ? rest : popped) : cu;
                    tr[j] = load_tri(sc.tris, ks[j]);
                }
                // the tests, each with the tmax the earlier ones left.  any-hit: the first accepted hit that is not the
                // excluded triangle (bvh.cuh:243); closest-hit: bvh.cuh:227-231 (t <= tmax), ties by closest_hit_wins
                bool occluded = false;
                int occ_j = 0;  // which of the tests found the occluder
#pragma unroll
                for (int j = 0; j < kTriPerStep; j++) {
                    if (act[j] && !occluded) {
                        float t, u, v;
                        const bool hit = tri_intersect(tr[j], o, d, tmax, t, u, v);
                        occluded = hit && is_any && ks[j] != tri;
                        occ_j = occluded ? j : occ_j;
                        bool better = hit && !is_any;
                        if (VERIFY) {
                            // an exact tie is the reference's tree order to decide: the ray is re-traced literally in the ADV block
                            // (the mark is the sign bit of hv; v >= 0 for an accepted hit), so which of the two stays until then
                            // does not matter -- no branch, no look at the caller order
                            const bool tie = better && t == tmax && tri >= 0;
                            v = __uint_as_float(__float_as_uint(v) | (tie ? 0x80000000u : 0u));
                        } else if (better && t == tmax && tri >= 0) {  // (RT_FLAG_WATERTIGHT: ties go to the larger caller index)
                            better = sc.order[(unsigned)ks[j]] > sc.order[(unsigned)tri];
                        }
                        tmax = better ? t : tmax;
                        hu = occluded ? 1.f : (better ? u : hu);
                        hv = better ? v : hv;
                        tri = better ? ks[j] : tri;
                    }
                }
                pend = occluded ? kEntryDone : pd;
                cur = occluded ? kEntryDone : cu;
#ifndef RT_DBG_NO_TRI_VERIFY  // (timing experiment only)
                // VERIFY: an occluder only counts if the reference's walk can see its triangle (2 % of the shadow rays get here;
                // ONE branch behind both tests: inside each test it cost the block's straight-line shape, 2 % of the frame).  An
                // occluder it cannot see -- ~1 in 10^7 -- says nothing about the rest of the ray: the ray ends here and is
                // re-traced through the reference's own tree in the finished-rays section (hu = 2 marks it)
                if (VERIFY && occluded) {
                    static_assert(kTriPerStep == 2, "the occluder is picked from two records");
                    Tri tq;
                    tq.p0 = occ_j ? tr[1].p0 : tr[0].p0;
                    tq.e1 = occ_j ? tr[1].e1 : tr[0].e1;
                    tq.e2 = occ_j ? tr[1].e2 : tr[0].e2;
                    tq.n = tq.p0;  // (not looked at)
                    if (!ref_visible(sc, o, d, tq, occ_j ? ks[1] : ks[0], vstat)) hu = 2.f;
                }
#endif
            }
#ifdef RT_TRACE_PROFILE
            pf[10] += __builtin_readcyclecounter() - pf_tt;
#endif
            RT_MARK("tri.end");
        };
        // Which of the two.  MAJORITY: the more popular block -- and when that is the node block, the triangle block right
        // behind it for the lanes that hold a leaf BY THEN (at least `tri_follow` of them): a lane that reached a leaf in
        // the node block has its triangles tested in this scheduling round instead of the next one.  Measured on C2:
        // +6.6 % at tri_follow = 1, +5.3 % at 12, +0.4 % at 40; the mirror image (a node block behind a triangle block) buys
        // nothing on top and loses 5 % alone.  Without MAJORITY: both blocks, for the lanes that wanted them at the top.
        if (!MAJORITY) {
            if (n_node > 0) node_block(want_node, n_node);
            if (n_tri > 0) tri_block(want_tri, n_tri);
        }
#ifndef RT_TRI_TWO_COPIES
        else {
            // (ONE copy of the triangle block in the code: the block behind a node block and the block on its own are the same
            // instructions for different lanes)
            const bool run_node = n_node > 0 && n_node >= n_tri;
            bool w = want_tri;
            int nw = n_tri;
            if (run_node) {
                node_block(want_node, n_node);
                w = tri_follow > 0 && trav && ((cur != kEntryDone && cur < 0) || (kSpeculate && pend != kEntryDone));
                nw = wave_count(w);
                nw = (tri_follow > 0 && nw >= tri_follow) ? nw : 0;
            }
            if (nw > 0) tri_block(w, nw);
        }
#else
        else if (n_node > 0 && n_node >= n_tri) {
            node_block(want_node, n_node);
            if (tri_follow > 0) {
                const bool w = trav && ((cur != kEntryDone && cur < 0) || (kSpeculate && pend != kEntryDone));
                const int nw = wave_count(w);
                if (nw >= tri_follow) tri_block(w, nw);
            }
        } else if (n_tri > 0) {
            tri_block(want_tri, n_tri);
        }
#endif
        // ---------------- finished rays
        RT_MARK("fin.begin");
        const bool fin = trav && cur == kEntryDone && (!kSpeculate || pend == kEntryDone);
#ifdef RT_TRACE_PROFILE
        const unsigned long long pf_tf = __builtin_readcyclecounter();
        pf_fin_lanes += wave_count(fin);
        pf_fin_iters += wave_ballot(fin) != 0 ? 1 : 0;
#endif
        if (VERIFY) {  // the shadow rays whose occluder the reference cannot see (triangle block): the literal walk decides
            const bool lit = fin && is_any && hu == 2.f;
            if (wave_ballot(lit)) {
                if (lit) {
                    atomicAdd(&vstat[V_LITERAL], 1ull);
                    float tm = tmax, no_v = 0.f;
                    int excluded = tri;
                    hu = 0.f;
                    reference_walk<true>(sc, o, d, tm, excluded, hu, no_v, stack, over, stack_cap);
                }
            }
        }
        n_deposit += wave_count((fin && is_any && hu == 0.f));
        if (fin) {
            if (is_any) {
                if (hu == 0.f && !debug_no_deposit)  // unoccluded: render.cuh:291-293
                    acc_add(acc, park[6 * kBlock], park[7 * kBlock], park[8 * kBlock]);
                // now the slot's path ray
                o = mk(park[0 * kBlock], park[1 * kBlock], park[2 * kBlock]);
                d = mk(park[3 * kBlock], park[4 * kBlock], park[5 * kBlock]);
                phase = PH_CLOSEST;
                inv = inv_dir(d);
                tmax = kFltMax;
                tri = -1;
                cur = 0;
                sp = 0;
            } else {
                // (tri, hu, hv, d) carry the hit to the ADV block; a path that missed, or has no bounce left (and is not
                // at bounce 0, where a hit light still emits: render.cuh:98-109), can only generate
                const int b = cold[0 * kBlock];
                phase = (SPLIT_GEN && (tri < 0 || (b >= ap_max_bounces && b > 0))) ? PH_GEN : PH_ADV;
            }
        }
#ifdef RT_TRACE_PROFILE
        pf_fin_cycles += __builtin_readcyclecounter() - pf_tf;
#endif
    }
#ifdef RT_TRACE_PROFILE
    if (prof && lane_id() == 0) { atomicAdd(&prof[17], pf_fin_cycles); atomicAdd(&prof[18], pf_fin_lanes); atomicAdd(&prof[19], pf_fin_iters); }
    if (prof && lane_id() == 0)
        { pf[11] = __builtin_readcyclecounter() - pf_t0; for (int k = 0; k < 16; k++) atomicAdd(&prof[k], pf[k]); atomicMax(&prof[13], pf[11]); atomicAdd(&prof[14], 1ull); atomicAdd(&prof[16], pf_gen_cycles);
          // per-wave record: where it ran and for how long
          unsigned long long *rec = prof + 24 + 4 * (size_t)((blockIdx.x * kBlock + threadIdx.x) >> 6);
          rec[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
          rec[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
          rec[2] = pf[11];
          rec[3] = pf[0] + pf[2] + pf[4]; }
#endif
    unsigned long long v[C_COUNT] = {n_gen, n_shade, n_traced, n_shadow, n_emit, n_deposit, n_rr, 0ull};
    row_add(rows, v);
}

// post_process_framebuffer (render.cuh:330-338): c = sqrt(c * (1/spp))
__global__ void k_post_process(float *fb, int n_values, float inv_spp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_values) fb[i] = sqrtf(fb[i] * inv_spp);
}

// fixed-point sums -> post-processed image: c = sqrt(float(sum * 2^-30) * (1/spp))
__global__ void k_post_process_fixed(const long long *__restrict__ sums, float *__restrict__ out, int n_values, float inv_spp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_values) out[i] = sqrtf((float)((double)sums[i] * (1.0 / 1073741824.0)) * inv_spp);
}

// rt_render_multi: dst += src over the raw sums of two shards (fp32 sums, or the int64 fixed-point sums of RT_FLAG_DETERMINISTIC)
__global__ void k_accumulate_f32(float *__restrict__ dst, const float *__restrict__ src, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
__global__ void k_accumulate_i64(long long *__restrict__ dst, const long long *__restrict__ src, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// ---- stage-level test kernels
__global__ void k_test_draw(DPools p, int n, int draws, uint32_t *__restrict__ state6, float *__restrict__ uni) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rs{p.rd(i), p.r0(i), p.r1(i), p.r2(i), p.r3(i), p.r4(i)};
    for (int k = 0; k < draws; k++) uni[(size_t)i * draws + k] = rng_uniform(rs);
    state6[6 * (size_t)i + 0] = rs.d;
    state6[6 * (size_t)i + 1] = rs.v0;
    state6[6 * (size_t)i + 2] = rs.v1;
    state6[6 * (size_t)i + 3] = rs.v2;
    state6[6 * (size_t)i + 4] = rs.v3;
    state6[6 * (size_t)i + 5] = rs.v4;
}
// ============================================================================ device BVH build (LBVH)
// a BVH build on the GPU.  Optional (RT_BVH_BUILDER=lbvh): a linear BVH --
// 30-bit Morton codes of the triangle centroids, sorted, binary radix tree by longest common prefix
// (Karras 2012), bottom-up box fit -- emitted in the same 2-wide record format the kernels walk.  It
// builds in about a millisecond but has no surface-area heuristic, so traversal is slower than through
// the host SAH tree; the image is the same (closest accepted triangle does not depend on the tree).
__device__ __forceinline__ unsigned morton_expand(unsigned v) {  // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__global__ void k_lbvh_keys(const float *__restrict__ verts, int n, int n_pad, float lox, float loy, float loz, float sx,
                            float sy, float sz, unsigned long long *__restrict__ keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    if (i >= n) {
        keys[i] = ~0ull;  // padding sorts last
        return;
    }
    const float *v = verts + 9 * (size_t)i;
    float cx = (fminf(v[0], fminf(v[3], v[6])) + fmaxf(v[0], fmaxf(v[3], v[6]))) * 0.5f;
    float cy = (fminf(v[1], fminf(v[4], v[7])) + fmaxf(v[1], fmaxf(v[4], v[7]))) * 0.5f;
    float cz = (fminf(v[2], fminf(v[5], v[8])) + fmaxf(v[2], fmaxf(v[5], v[8]))) * 0.5f;
    unsigned qx = (unsigned)fminf(fmaxf((cx - lox) * sx, 0.f), 1023.f);
    unsigned qy = (unsigned)fminf(fmaxf((cy - loy) * sy, 0.f), 1023.f);
    unsigned qz = (unsigned)fminf(fmaxf((cz - loz) * sz, 0.f), 1023.f);
    unsigned code = (morton_expand(qx) << 2) | (morton_expand(qy) << 1) | morton_expand(qz);
    keys[i] = ((unsigned long long)code << 32) | (unsigned)i;  // the index makes every key unique
}
__global__ void k_bitonic_step(unsigned long long *__restrict__ keys, int n_pad, int j, int k) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    int partner = i ^ j;
    if (partner > i) {
        unsigned long long a = keys[i], b = keys[partner];
        bool ascending = (i & k) == 0;
        if ((a > b) == ascending) {
            keys[i] = b;
            keys[partner] = a;
        }
    }
}
__device__ __forceinline__ int lbvh_delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}
// child encoding: internal node k -> k, leaf k (sorted position) -> ~k
__global__ void k_lbvh_hierarchy(const unsigned long long *__restrict__ keys, int n, int *__restrict__ left,
                                 int *__restrict__ right, int *__restrict__ parent_int, int *__restrict__ parent_leaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = lbvh_delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int lc = (lo == gamma) ? ~gamma : gamma;
    int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    if (lc < 0) parent_leaf[~lc] = i; else parent_int[lc] = i;
    if (rc < 0) parent_leaf[~rc] = i; else parent_int[rc] = i;
    if (i == 0) parent_int[0] = -1;
}
__device__ __forceinline__ void lbvh_leaf_box(const float *verts, const unsigned long long *keys, int pos, float *b) {
    const float *v = verts + 9 * (size_t)(unsigned)(keys[pos] & 0xffffffffu);
    for (int a = 0; a < 3; a++) {
        b[a] = fminf(v[a], fminf(v[3 + a], v[6 + a]));
        b[3 + a] = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
    }
}
// bottom-up: the second thread to arrive at an internal node owns it (its sibling subtree is complete)
__global__ void k_lbvh_fit(const float *__restrict__ verts, const unsigned long long *__restrict__ keys, int n,
                           const int *__restrict__ left, const int *__restrict__ right, const int *__restrict__ parent_int,
                           const int *__restrict__ parent_leaf, float *boxes, int *depth, int *arrivals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cur = parent_leaf[i];
    for (int guard = 0; cur >= 0 && guard < 4096; guard++) {  // (a radix tree over 64-bit keys is at most 64 deep)
        __threadfence();  // publish what this thread wrote below before announcing arrival
        if (atomicAdd(&arrivals[cur], 1) == 0) return;
        __threadfence();  // second arrival: the sibling's box and depth are visible from here on
        float b[6] = {kFltMax, kFltMax, kFltMax, -kFltMax, -kFltMax, -kFltMax};
        int dep = 0;
        const int ch[2] = {left[cur], right[cur]};
        for (int c = 0; c < 2; c++) {
            float cb[6];
            int cd = 0;
            if (ch[c] < 0) {
                lbvh_leaf_box(verts, keys, ~ch[c], cb);
            } else {
                const volatile float *vb = boxes + 6 * (size_t)ch[c];
                for (int a = 0; a < 6; a++) cb[a] = vb[a];
                cd = ((const volatile int *)depth)[ch[c]];
            }
            for (int a = 0; a < 3; a++) {
                b[a] = fminf(b[a], cb[a]);
                b[3 + a] = fmaxf(b[3 + a], cb[3 + a]);
            }
            dep = max(dep, cd);
        }
        for (int a = 0; a < 6; a++) boxes[6 * (size_t)cur + a] = b[a];
        depth[cur] = dep + 1;
        cur = parent_int[cur];
    }
}
__device__ __forceinline__ float lbvh_pad(float v, int dir) {  // 2 ulps outward, as the host builder pads
    v = nextafterf(v, dir < 0 ? -kFltMax : kFltMax);
    return nextafterf(v, dir < 0 ? -kFltMax : kFltMax);
}
__global__ void k_lbvh_emit(const float *__restrict__ verts, const unsigned long long *__restrict__ keys, int n,
                            const int *__restrict__ left, const int *__restrict__ right, const float *__restrict__ boxes,
                            float *__restrict__ pairs, int *__restrict__ order) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = (int)(unsigned)(keys[i] & 0xffffffffu);
    if (i >= n - 1) return;
    float *rec = pairs + 16 * (size_t)i;
    const int ch[2] = {left[i], right[i]};
    for (int c = 0; c < 2; c++) {
        float cb[6];
        int link;
        if (ch[c] < 0) {
            lbvh_leaf_box(verts, keys, ~ch[c], cb);
            link = ~(((~ch[c]) << 3) | 1);  // leaf reference: one triangle at sorted position
        } else {
            for (int a = 0; a < 6; a++) cb[a] = boxes[6 * (size_t)ch[c] + a];
            link = ch[c];
        }
        for (int a = 0; a < 3; a++) {
            rec[6 * c + a] = lbvh_pad(cb[a], -1);
            rec[6 * c + 3 + a] = lbvh_pad(cb[3 + a], +1);
        }
        rec[12 + c] = __int_as_float(link);
    }
    rec[14] = 0.f;
    rec[15] = 0.f;
}

// ============================================================================ host side
struct rt_scene {
    int device = 0;
    int n_tris = 0, n_nodes = 0, max_depth = 0, stack_bound = 1, n_leaves = 0, n_lights = 0, n_mats = 0;
    float4 *d_nodes = nullptr;
    bool wide = false;  // node records: 4-wide (two pair-style records per node, rtbvh::Result::quads) or 2-wide (rtbvh::Pair)
    // 4-wide records are padded for the ray origins that will be traced (rt_bvh.h, pad_quads_for_origins): the builder's
    // records, the radius the device copy is padded for at the moment, and a lock for the (rare) re-padding
    std::vector<rtbvh::Pair> h_quads;
    mutable float origin_radius[3] = {0.f, 0.f, 0.f};
    mutable std::mutex pad_mutex;
    bool top_prefix = true;  // the first records are the top of the tree in level order (host builder)
    double build_seconds = 0.0;  // BVH build time (host wall clock, or device events for the LBVH)
    int builder = 0;             // 0 host SAH, 1 device LBVH
    float4 *d_tris = nullptr;
    int2 *d_tri_info = nullptr;
    float4 *d_tri_shade = nullptr;
    Material *d_mats = nullptr;
    Light *d_lights = nullptr;
    float *d_tables = nullptr;    // shading tables (see DScene)
    int tab_dwords = 0;
    int *d_order = nullptr;       // leaf order -> original
    std::vector<int> h_order;     // leaf order -> original
    std::vector<int> h_inverse;   // original -> leaf order
    // RT_FLAG_REFERENCE_WALK: the reference's own tree (rt_ref_tree.h), built and uploaded by the first render that asks
    // for it (ensure_ref_tree) from the caller's triangles kept here
    std::vector<float> h_tri9;
    // rt_render_multi: what a replica of this scene on another device is created from, and the replicas made so far
    std::vector<int32_t> h_tri_material, h_tri_light;
    std::vector<rt_material> h_materials;
    std::vector<rt_light> h_lights;
    mutable std::mutex replica_mutex;
    mutable std::vector<rt_scene *> replicas;  // owned; at most one per device
    mutable std::mutex ref_mutex;
    mutable bool ref_ready = false;
    mutable float4 *d_ref_nodes = nullptr;
    mutable int *d_ref_prims = nullptr;
    mutable int *d_ref_leaf_of = nullptr;  // leaf-order triangle index -> node of its leaf in the reference's tree (ref_visible)
    mutable int *d_ref_parent = nullptr;   // node -> parent node (root: -1)
    mutable int ref_nodes_count = 0, ref_depth = 0;
    mutable bool ref_root_leaf = true;
    mutable double build_seconds_ref = 0.0;  // host time of the reference-tree build + upload (one-off, first render that needs it)
    rt_scene() = default;
    rt_scene(const rt_scene &) = delete;
    rt_scene &operator=(const rt_scene &) = delete;
    ~rt_scene() {  // (every early return of rt_scene_create goes through here: nothing leaks on an error path)
        for (rt_scene *r : replicas) delete r;
        (void)hipFree(d_nodes);
        (void)hipFree(d_tris);
        (void)hipFree(d_tri_info);
        (void)hipFree(d_tri_shade);
        (void)hipFree(d_mats);
        (void)hipFree(d_lights);
        (void)hipFree(d_order);
        (void)hipFree(d_tables);
        (void)hipFree(d_ref_nodes);
        (void)hipFree(d_ref_prims);
        (void)hipFree(d_ref_leaf_of);
        (void)hipFree(d_ref_parent);
    }
    DScene dev() const {
        DScene s;
        s.nodes = d_nodes;
        s.tris = d_tris;
        s.tri_info = d_tri_info;
        s.tri_shade = d_tri_shade;
        s.order = d_order;
        s.mats = d_mats;
        s.lights = d_lights;
        s.num_lights = n_lights;
        s.num_mats = n_mats;
        s.tables = d_tables;
        s.tab_dwords = tab_dwords;
        s.ref_nodes = d_ref_nodes;
        s.ref_prims = d_ref_prims;
        s.ref_n_prims = ref_ready ? n_tris : 0;
        s.ref_leaf_of = d_ref_leaf_of;
        s.ref_parent = d_ref_parent;
        s.ref_root_leaf = ref_root_leaf ? 1 : 0;
        return s;
    }
};

namespace {

// ---- XORWOW host pieces: seed scramble and the 2^67 jump matrices J^(2^k)
Rng xorwow_seed(uint64_t seed) {  // curand_init's scramble (curand_kernel.h)
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    Rng st;
    st.d = 6615241u + t1 + t0;
    st.v0 = 123456789u + t0;
    st.v1 = 362436069u ^ t0;
    st.v2 = 521288629u + t1;
    st.v3 = 88675123u ^ t1;
    st.v4 = 5783321u + t0;
    return st;
}
typedef uint32_t Mat160[160][5];
void mat160_apply(const Mat160 &m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; w++)
        for (int b = 0; b < 32; b++)
            if (in[w] & (1u << b))
                for (int k = 0; k < 5; k++) r[k] ^= m[w * 32 + b][k];
    memcpy(out, r, sizeof(r));
}
void mat160_square(Mat160 &m) {
    static Mat160 tmp;
    for (int i = 0; i < 160; i++) mat160_apply(m, m[i], tmp[i]);
    memcpy(m, tmp, sizeof(Mat160));
}
// host table: 20 matrices J^(2^k), J = (one xorwow step)^(2^67)
const std::vector<uint32_t> &jump_powers() {
    static std::vector<uint32_t> table;
    static std::once_flag once;
    std::call_once(once, [] {
        static Mat160 a;
        for (int w = 0; w < 5; w++)
            for (int b = 0; b < 32; b++) {
                uint32_t v[5] = {0, 0, 0, 0, 0};
                v[w] = 1u << b;
                uint32_t t = v[0] ^ (v[0] >> 2);
                v[0] = v[1];
                v[1] = v[2];
                v[2] = v[3];
                v[3] = v[4];
                v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
                memcpy(a[w * 32 + b], v, sizeof(v));
            }
        for (int s = 0; s < 67; s++) mat160_square(a);
        table.resize((size_t)20 * 160 * 5);
        for (int k = 0; k < 20; k++) {
            memcpy(table.data() + (size_t)k * 800, a, sizeof(Mat160));
            mat160_square(a);
        }
    });
    return table;
}

// Structural check of 2-wide records before they are uploaded (a malformed tree would hang the GPU):
// every record reachable from the root exactly once, every triangle position in exactly one leaf.
bool validate_pairs(const std::vector<rtbvh::Pair> &pairs, int n_tris) {
    const int np = (int)pairs.size();
    if (np == 0) return false;
    std::vector<char> seen_pair(np, 0), seen_tri((size_t)std::max(n_tris, 1), 0);
    std::vector<int> todo{0};
    seen_pair[0] = 1;
    int visited = 0, tris = 0;
    while (!todo.empty()) {
        int pi = todo.back();
        todo.pop_back();
        visited++;
        const int links[2] = {pairs[pi].llink, pairs[pi].rlink};
        for (int l : links) {
            if (l == rtbvh::kNoChild) continue;
            if (l >= 0) {
                if (l >= np || seen_pair[l]) return false;
                seen_pair[l] = 1;
                todo.push_back(l);
            } else {
                int ref = ~l, first = ref >> 3, count = ref & 7;
                if (count <= 0 || first < 0 || first + count > n_tris) return false;
                for (int k = first; k < first + count; k++) {
                    if (seen_tri[k]) return false;
                    seen_tri[k] = 1;
                    tris++;
                }
            }
        }
    }
    return visited == np && tris == n_tris;
}

// The same for the 4-wide format (two consecutive records per node; inner links are even record indices), plus the
// invariant the kernels' box test rests on: a child is absent (link kNoChild) if and only if its box is all +inf -- the
// one-comparison slab test of inner_step<true> never looks at links.
// Device layout.  2-wide: a 64-byte record with the two children's bounds INTERLEAVED --
//   (l.lo.x, r.lo.x, l.lo.y, r.lo.y | l.lo.z, r.lo.z, l.hi.x, r.hi.x | l.hi.y, r.hi.y, l.hi.z, r.hi.z | llink, rlink, spare, spare)
// -- so that every (left, right) pair of bounds arrives in an aligned register pair and the slab arithmetic of both children
// runs as packed fp32, see inner_step.  4-wide: a node (two builder records: children 0, 1 | children 2, 3, rt_bvh.h `quads`,
// padded for ray origins within `radius`: pad_quads_for_origins) is 128 bytes laid out BY PLANE -- see below and inner_step:
// a node step loads seven 16-byte words (a divergent wave-wide load occupies the CU's texture addresser for about a cycle
// per active lane: profiles/r05_gather_rate.txt) and picks near and far planes by address instead of by min / max.
int upload_node_records(const rt_scene *sc, const std::vector<rtbvh::Pair> &base, const float radius[3]) {
    std::vector<rtbvh::Pair> padded;
    const std::vector<rtbvh::Pair> *recs = &base;
    if (sc->wide) {
        rtbvh::pad_quads_for_origins(base, radius, padded);
        recs = &padded;
    }
    if ((size_t)sc->n_nodes != recs->size()) return fail("upload_node_records: record count changed");
    std::vector<float> inter(16 * recs->size());
    if (!sc->wide) {
        for (size_t k = 0; k < recs->size(); k++) {
            const rtbvh::Pair &pr = (*recs)[k];
            float *r = &inter[16 * k];
            for (int a = 0; a < 6; a++) {
                r[2 * a] = pr.lbox[a];
                r[2 * a + 1] = pr.rbox[a];
            }
            memcpy(&r[12], &pr.llink, 4);
            memcpy(&r[13], &pr.rlink, 4);
            r[14] = r[15] = 0.f;
        }
    } else {
        // 4-wide node j = builder records 2j (children 0, 1) and 2j + 1 (children 2, 3) -> 128 bytes BY PLANE:
        //   word 2a: the four children's lower bounds of axis a, word 2a + 1: their upper bounds (a = x, y, z), word 6: the
        //   four links, word 7: spare
        for (size_t j = 0; 2 * j + 1 < recs->size(); j++) {
            const rtbvh::Pair &p0 = (*recs)[2 * j], &p1 = (*recs)[2 * j + 1];
            const float *box[4] = {p0.lbox, p0.rbox, p1.lbox, p1.rbox};
            const int32_t link[4] = {p0.llink, p0.rlink, p1.llink, p1.rlink};
            float *r = &inter[32 * j];
            for (int a = 0; a < 3; a++)
                for (int c = 0; c < 4; c++) {
                    r[8 * a + c] = box[c][a];
                    r[8 * a + 4 + c] = box[c][3 + a];
                }
            memcpy(&r[24], link, 16);
            r[28] = r[29] = r[30] = r[31] = 0.f;
        }
    }
    HIP_TRY(hipMemcpy(sc->d_nodes, inter.data(), 64 * (size_t)sc->n_nodes, hipMemcpyHostToDevice));
    for (int a = 0; a < 3; a++) sc->origin_radius[a] = radius[a];
    return 0;
}
// Before rays are traced whose origins may lie outside the radius the 4-wide records are padded for (a camera outside the
// scene's bounds; the rays of the test hooks): re-pad, generously, and upload.  Renders of the same scene that are in flight
// on other streams read a mix of the old and the new bounds meanwhile -- both are conservative for THEIR rays.
int ensure_origin_radius(const rt_scene *sc, const float need[3]) {
    if (!sc->wide) return 0;
    std::lock_guard<std::mutex> lock(sc->pad_mutex);
    bool grow = false;
    float radius[3];
    for (int a = 0; a < 3; a++) {
        const float want = std::isfinite(need[a]) ? std::fabs(need[a]) * 1.001f : 0.f;  // (a non-finite origin hits nothing anyway)
        grow = grow || want > sc->origin_radius[a];
        radius[a] = want > sc->origin_radius[a] ? 2.f * want : sc->origin_radius[a];
    }
    if (!grow) return 0;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != sc->device) HIP_TRY(hipSetDevice(sc->device));
    const int rc = upload_node_records(sc, sc->h_quads, radius);
    if (dev != sc->device) HIP_TRY(hipSetDevice(dev));
    return rc;
}

bool validate_quads(const std::vector<rtbvh::Pair> &quads, int n_tris) {
    const int nr = (int)quads.size();
    if (nr < 2 || (nr & 1)) return false;
    std::vector<char> seen_node((size_t)nr / 2, 0), seen_tri((size_t)std::max(n_tris, 1), 0);
    std::vector<int> todo{0};
    seen_node[0] = 1;
    int visited = 0, tris = 0;
    while (!todo.empty()) {
        const int rec = todo.back();
        todo.pop_back();
        visited++;
        for (int k = 0; k < 4; k++) {
            const rtbvh::Pair &p = quads[(size_t)rec + (k >> 1)];
            const int l = (k & 1) ? p.rlink : p.llink;
            const float *b = (k & 1) ? p.rbox : p.lbox;
            bool all_inf = true, finite = true;
            for (int a = 0; a < 6; a++) {
                all_inf = all_inf && b[a] == INFINITY;
                finite = finite && std::isfinite(b[a]);
            }
            if (l == rtbvh::kNoChild) {
                if (!all_inf) return false;
                continue;
            }
            if (!finite || b[0] > b[3] || b[1] > b[4] || b[2] > b[5]) return false;
            if (l >= 0) {
                if ((l & 1) || l >= nr || seen_node[l / 2]) return false;
                seen_node[l / 2] = 1;
                todo.push_back(l);
            } else {
                const int ref = ~l, first = ref >> 3, count = ref & 7;
                if (count <= 0 || first < 0 || first + count > n_tris) return false;
                for (int t = first; t < first + count; t++) {
                    if (seen_tri[t]) return false;
                    seen_tri[t] = 1;
                    tris++;
                }
            }
        }
    }
    return visited == nr / 2 && tris == n_tris;
}

// RT_FLAG_REFERENCE_WALK: build the reference's tree from the caller's triangles, check its structure (a malformed tree
// would hang the walk: every node reached exactly once, children adjacent, every primitive position in exactly one
// leaf, depth within the walk's private stack) and upload it.  Once per scene, on the scene's device.
int ensure_ref_tree(const rt_scene *scene) {
    std::lock_guard<std::mutex> lock(scene->ref_mutex);
    if (scene->ref_ready) return 0;
    const auto t_begin = std::chrono::steady_clock::now();
    const int n = scene->n_tris;
    if ((int)scene->h_tri9.size() != 9 * n) return fail("RT_FLAG_REFERENCE_WALK: the scene holds no triangle copy");
    const rtref::Tree t = rtref::build(scene->h_tri9.data(), n);
    const int nn = (int)t.nodes.size();
    if (n > 0) {
        std::vector<char> seen_node((size_t)nn, 0), seen_prim((size_t)n, 0);
        std::vector<std::pair<int, int>> todo{{0, 0}};  // (node, depth)
        int visited = 0, prims = 0;
        seen_node[0] = 1;
        while (!todo.empty()) {
            const auto [k, dep] = todo.back();
            todo.pop_back();
            visited++;
            const rtref::Node &nd = t.nodes[(size_t)k];
            if (nd.count > 0) {
                if (nd.link < 0 || nd.link + nd.count > n) return fail("RT_FLAG_REFERENCE_WALK: malformed leaf");
                for (int i = nd.link; i < nd.link + nd.count; i++) {
                    if (seen_prim[i]) return fail("RT_FLAG_REFERENCE_WALK: primitive in two leaves");
                    seen_prim[i] = 1;
                    prims++;
                }
            } else {
                if (nd.count < 0 || nd.link <= 0 || nd.link + 1 >= nn || seen_node[nd.link] || seen_node[nd.link + 1] || dep >= rtref::kMaxDepth)
                    return fail("RT_FLAG_REFERENCE_WALK: malformed inner node");
                seen_node[nd.link] = seen_node[nd.link + 1] = 1;
                todo.push_back({nd.link, dep + 1});
                todo.push_back({nd.link + 1, dep + 1});
            }
        }
        if (visited != nn || prims != n) return fail("RT_FLAG_REFERENCE_WALK: tree does not cover the scene");
        for (int i = 0; i < n; i++)
            if (t.prims[i] < 0 || t.prims[i] >= n) return fail("RT_FLAG_REFERENCE_WALK: bad primitive order");
    }
    std::vector<int> prim_leaf((size_t)std::max(n, 1), 0);  // reference primitive position -> this scene's leaf-order index
    for (int i = 0; i < n; i++) prim_leaf[i] = scene->h_inverse[t.prims[i]];
    // what ref_visible reads: the leaf of every triangle (leaf-order index -> node) and the way up from there
    std::vector<int> leaf_of((size_t)std::max(n, 1), 0), parent((size_t)std::max(nn, 1), -1);
    for (int k = 0; k < nn; k++) {
        const rtref::Node &nd = t.nodes[(size_t)k];
        if (nd.count > 0) {
            for (int i = nd.link; i < nd.link + nd.count; i++) leaf_of[(size_t)prim_leaf[(size_t)i]] = k;
        } else if (n > 0) {
            parent[(size_t)nd.link] = parent[(size_t)nd.link + 1] = k;
        }
    }
    float4 *dn = nullptr;
    int *dp = nullptr, *dl = nullptr, *dpar = nullptr;
    if (hipMalloc((void **)&dn, sizeof(rtref::Node) * (size_t)std::max(nn, 1)) != hipSuccess ||
        hipMalloc((void **)&dp, sizeof(int) * prim_leaf.size()) != hipSuccess ||
        hipMalloc((void **)&dl, sizeof(int) * leaf_of.size()) != hipSuccess ||
        hipMalloc((void **)&dpar, sizeof(int) * parent.size()) != hipSuccess ||
        hipMemcpy(dn, t.nodes.data(), sizeof(rtref::Node) * (size_t)nn, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dp, prim_leaf.data(), sizeof(int) * prim_leaf.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dl, leaf_of.data(), sizeof(int) * leaf_of.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dpar, parent.data(), sizeof(int) * parent.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(dn);
        (void)hipFree(dp);
        (void)hipFree(dl);
        (void)hipFree(dpar);
        return fail("reference tree: device allocation or upload failed");
    }
    scene->d_ref_nodes = dn;
    scene->d_ref_prims = dp;
    scene->d_ref_leaf_of = dl;
    scene->d_ref_parent = dpar;
    scene->ref_root_leaf = n == 0 || t.nodes[0].count > 0;
    scene->ref_nodes_count = nn;
    scene->ref_depth = t.depth;
    scene->build_seconds_ref = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    scene->ref_ready = true;
    return 0;
}

// Device temporaries and events of one host call: released on EVERY return path (HIP_TRY returns early on errors)
struct DevScope {
    std::vector<void *> ptrs;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~DevScope() {
        for (void *q : ptrs) (void)hipFree(q);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    template <typename T>
    int alloc(T *&ptr, size_t count) {
        void *raw = nullptr;
        HIP_TRY(hipMalloc(&raw, std::max<size_t>(count, 1) * sizeof(T)));
        ptrs.push_back(raw);
        ptr = (T *)raw;
        return 0;
    }
};
// Device LBVH build: returns the pair records and the leaf order on the host (the caller uploads them
// like the host builder's output).  n >= 2.
int build_lbvh_device(const float *verts_host, int n, std::vector<rtbvh::Pair> &pairs, std::vector<int32_t> &order,
                      int &depth, double &seconds) {
    float lo[3] = {kFltMax, kFltMax, kFltMax}, hi[3] = {-kFltMax, -kFltMax, -kFltMax};
    for (size_t i = 0; i < (size_t)n * 3; i++)
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], verts_host[3 * i + a]);
            hi[a] = std::max(hi[a], verts_host[3 * i + a]);
        }
    float sc3[3];
    for (int a = 0; a < 3; a++) sc3[a] = hi[a] > lo[a] ? 1024.f / (hi[a] - lo[a]) : 0.f;
    int n_pad = 1;
    while (n_pad < n) n_pad <<= 1;
    float *d_verts = nullptr, *d_boxes = nullptr, *d_pairs = nullptr;
    unsigned long long *d_keys = nullptr;
    int *d_left = nullptr, *d_right = nullptr, *d_pi = nullptr, *d_pl = nullptr, *d_depth = nullptr, *d_arr = nullptr, *d_order = nullptr;
    DevScope tmp;  // the eleven temporaries and both events go away on every return path
    if (tmp.alloc(d_verts, 9 * (size_t)n) || tmp.alloc(d_keys, (size_t)n_pad) || tmp.alloc(d_left, (size_t)n) ||
        tmp.alloc(d_right, (size_t)n) || tmp.alloc(d_pi, (size_t)n) || tmp.alloc(d_pl, (size_t)n) || tmp.alloc(d_depth, (size_t)n) ||
        tmp.alloc(d_arr, (size_t)n) || tmp.alloc(d_order, (size_t)n) || tmp.alloc(d_boxes, 6 * (size_t)n) || tmp.alloc(d_pairs, 16 * (size_t)n))
        return 1;
    HIP_TRY(hipMemcpy(d_verts, verts_host, sizeof(float) * 9 * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_arr, 0, sizeof(int) * (size_t)n));
    HIP_TRY(hipMemset(d_depth, 0, sizeof(int) * (size_t)n));
    HIP_TRY(hipEventCreate(&tmp.e0));
    HIP_TRY(hipEventCreate(&tmp.e1));
    const hipEvent_t e0 = tmp.e0, e1 = tmp.e1;
    HIP_TRY(hipEventRecord(e0, nullptr));
    const dim3 blk(256);
    hipLaunchKernelGGL(k_lbvh_keys, dim3((n_pad + 255) / 256), blk, 0, nullptr, d_verts, n, n_pad, lo[0], lo[1], lo[2], sc3[0],
                       sc3[1], sc3[2], d_keys);
    for (int k2 = 2; k2 <= n_pad; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1)
            hipLaunchKernelGGL(k_bitonic_step, dim3((n_pad + 255) / 256), blk, 0, nullptr, d_keys, n_pad, j, k2);
    hipLaunchKernelGGL(k_lbvh_hierarchy, dim3((n + 255) / 256), blk, 0, nullptr, d_keys, n, d_left, d_right, d_pi, d_pl);
    hipLaunchKernelGGL(k_lbvh_fit, dim3((n + 255) / 256), blk, 0, nullptr, d_verts, d_keys, n, d_left, d_right, d_pi, d_pl,
                       d_boxes, d_depth, d_arr);
    hipLaunchKernelGGL(k_lbvh_emit, dim3((n + 255) / 256), blk, 0, nullptr, d_verts, d_keys, n, d_left, d_right, d_boxes,
                       d_pairs, d_order);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    seconds = ms * 1e-3;
    pairs.resize((size_t)n - 1);
    order.resize((size_t)n);
    HIP_TRY(hipMemcpy(pairs.data(), d_pairs, sizeof(rtbvh::Pair) * (size_t)(n - 1), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(order.data(), d_order, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&depth, d_depth, sizeof(int), hipMemcpyDeviceToHost));  // depth of the root
    return 0;
}

// ---- per-device render context: pools are allocated once per (device, n) and reused
struct Context {
    int device = -1;
    int n = 0;
    int lane = 0;  // concurrent sub-shards of one render use separate contexts (and streams)
    DPools pools{};
    std::vector<void *> allocs;
    DCounters *d_ctr = nullptr;
    DCounters *h_ctr = nullptr;  // pinned ring of snapshots
    DWaveRow *d_rows = nullptr;  // one row per wave of the stage grid
    int n_rows = 0;
    uint32_t *d_jump = nullptr;
    hipEvent_t ev_ring[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
    // RNG states are a pure function of (seed, slot range): keep them cached across renders
    uint64_t rng_seed = 0;
    int rng_lo = -1;
    bool rng_valid = false;
    uint32_t *rng_backup = nullptr;  // 6 x n words
    std::vector<hipEvent_t> timing_events;
    unsigned int *d_lock = nullptr;  // mat() events per lockstep round of the final generation (lock_cap words, grown on demand)
    int lock_cap = 0;
    int *d_over = nullptr;  // overflow part of the traversal stacks of this context's grids (ensure_overflow)
    int over_levels = 0;
    std::mutex busy;  // a context (pools, counters, events) serves one render at a time
    Context() = default;
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ~Context() {  // (rt_shutdown, or a context that failed half-way through get_context): on the device it lives on
        int saved = 0;
        const bool hop = device >= 0 && hipGetDevice(&saved) == hipSuccess && saved != device && hipSetDevice(device) == hipSuccess;
        for (void *q : allocs) (void)hipFree(q);
        (void)hipFree(d_lock);
        (void)hipFree(d_over);
        if (h_ctr) (void)hipHostFree(h_ctr);
        for (hipEvent_t e : ev_ring) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_a, ev_b, ev_c}) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : timing_events) (void)hipEventDestroy(e);
        if (hop) (void)hipSetDevice(saved);
    }
};
std::mutex g_ctx_mutex;
std::vector<std::unique_ptr<Context>> g_contexts;

// Device buffers of the calls that take HOST output (rt_render, rt_render_multi): raw sums, staging, the post-processed image.
// Kept per (device, slot) and reused from call to call -- a hipMalloc / hipFree pair per frame cost ~0.3 ms and a device
// synchronisation each (round 4 allocated them per call); released by rt_shutdown.  The buffers of ONE device serve one call at
// a time (g_dev_busy[device]); calls on different devices -- one host thread per GPU -- do not wait for each other.
struct OutBuffer {
    int device = -1, slot = 0;
    void *ptr = nullptr;
    size_t bytes = 0;
};
constexpr int kMaxDevices = 64;
std::mutex g_dev_busy[kMaxDevices];
std::mutex g_out_mutex;  // the list below
std::vector<OutBuffer> g_out_buffers;
// (caller holds g_dev_busy[device]; the current device must be `device`)
void *out_buffer(int device, int slot, size_t bytes) {
    std::lock_guard<std::mutex> list_lock(g_out_mutex);
    for (OutBuffer &b : g_out_buffers)
        if (b.device == device && b.slot == slot) {
            if (b.bytes >= bytes) return b.ptr;
            (void)hipFree(b.ptr);
            b.ptr = nullptr;
            b.bytes = 0;
            if (hipMalloc(&b.ptr, bytes) != hipSuccess) return nullptr;
            b.bytes = bytes;
            return b.ptr;
        }
    OutBuffer b;
    b.device = device;
    b.slot = slot;
    if (hipMalloc(&b.ptr, bytes) != hipSuccess) return nullptr;
    b.bytes = bytes;
    g_out_buffers.push_back(b);
    return b.ptr;
}

template <typename T>
int dev_alloc(Context &c, T *&ptr, size_t count) {
    void *raw = nullptr;
    HIP_TRY(hipMalloc(&raw, count * sizeof(T)));
    c.allocs.push_back(raw);
    ptr = (T *)raw;
    return 0;
}

int get_context(int n, int lane, Context **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    for (auto &c : g_contexts)
        if (c->device == dev && c->n == n && c->lane == lane) {
            *out = c.get();
            return 0;
        }
    auto c = std::make_unique<Context>();
    c->device = dev;
    c->n = n;
    c->lane = lane;
    DPools &p = c->pools;
    p.n = n;
    if (dev_alloc(*c, p.base, (size_t)A_COUNT * n)) return 1;
    if (dev_alloc(*c, c->rng_backup, (size_t)6 * n)) return 1;
    if (dev_alloc(*c, c->d_ctr, 1)) return 1;
    // one counter row per wave of the largest grid this context launches: k_advance's (n / 64 waves); the RT_HALF_WAVES
    // experiment launches k_paths with twice its usual waves, which on a small shard can exceed that
    c->n_rows = (knob("RT_HALF_WAVES") ? 2 : 1) * ((n + kBlock - 1) / kBlock) * (kBlock / 64);
    if (dev_alloc(*c, c->d_rows, (size_t)c->n_rows)) return 1;
    if (dev_alloc(*c, c->d_jump, (size_t)20 * 800)) return 1;
    HIP_TRY(hipMemcpy(c->d_jump, jump_powers().data(), sizeof(uint32_t) * 20 * 800, hipMemcpyHostToDevice));
    HIP_TRY(hipHostMalloc((void **)&c->h_ctr, sizeof(DCounters) * 4, hipHostMallocDefault));
    for (auto &e : c->ev_ring) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreate(&c->ev_a));
    HIP_TRY(hipEventCreate(&c->ev_b));
    HIP_TRY(hipEventCreate(&c->ev_c));
    *out = c.get();
    g_contexts.push_back(std::move(c));
    return 0;
}

int grid_for(int n) { return (n + kBlock - 1) / kBlock; }

// Entries of a lane's traversal stack kept in LDS (the rest of scene->stack_bound goes to the global overflow column).
// RT_STACK_CAP lowers it: a test knob that drives every traversal through the overflow path.
static int lds_stack_cap(const rt_scene *scene, int limit) {
    int cap = std::min(limit, std::max(1, scene->stack_bound));
    if (const char *e = knob("RT_STACK_CAP")) cap = std::max(1, std::min(cap, atoi(e)));
    return cap;
}

// launches k_advance<LDS tables?, material-sorted?> for one round (uses grid, block, sc, c, cam, ap, lds_tables of the caller)
static bool sort_shade() {
    const char *e = knob("RT_SORT_SHADE");
    return e ? atoi(e) != 0 : false;  // (measured on C2's round pipeline: 535 ms sorted, 517 ms in slot order -- see k_advance)
}
#define RT_LAUNCH_ADVANCE(stream, fbptr)                                                                                         \
    do {                                                                                                                         \
        if (lds_tables && sort_shade()) hipLaunchKernelGGL((k_advance<true, true>), grid, block, 0, stream, sc, c.pools, cam, ap, fbptr, c.d_ctr, c.d_rows, c.d_lock);   \
        else if (lds_tables) hipLaunchKernelGGL((k_advance<true, false>), grid, block, 0, stream, sc, c.pools, cam, ap, fbptr, c.d_ctr, c.d_rows, c.d_lock);             \
        else if (sort_shade()) hipLaunchKernelGGL((k_advance<false, true>), grid, block, 0, stream, sc, c.pools, cam, ap, fbptr, c.d_ctr, c.d_rows, c.d_lock);           \
        else hipLaunchKernelGGL((k_advance<false, false>), grid, block, 0, stream, sc, c.pools, cam, ap, fbptr, c.d_ctr, c.d_rows, c.d_lock);                            \
    } while (0)

// launches k_trace<MODE, wide?> -- the node format is a property of the scene
#define RT_LAUNCH_TRACE(MODE, wide, grid, lds, stream, ...)                                                \
    do {                                                                                                   \
        if (wide) hipLaunchKernelGGL((k_trace<MODE, true>), grid, dim3(kBlock), lds, stream, __VA_ARGS__);   \
        else hipLaunchKernelGGL((k_trace<MODE, false>), grid, dim3(kBlock), lds, stream, __VA_ARGS__);       \
    } while (0)
// ... or, with RT_FLAG_REFERENCE_WALK, the build that walks the reference's tree (the node format does not matter then);
// `verify`: the default build -- the product's walk with the reference's decisions (ref_visible); neither: RT_FLAG_WATERTIGHT
#define RT_LAUNCH_TRACE_REF(MODE, literal, verify, wide, grid, lds, stream, ...)                                          \
    do {                                                                                                                  \
        if (literal) hipLaunchKernelGGL((k_trace<MODE, false, 8, true>), grid, dim3(kBlock), lds, stream, __VA_ARGS__);     \
        else if ((verify) && (wide)) hipLaunchKernelGGL((k_trace<MODE, true, 8, false, true>), grid, dim3(kBlock), lds, stream, __VA_ARGS__); \
        else if (verify) hipLaunchKernelGGL((k_trace<MODE, false, 8, false, true>), grid, dim3(kBlock), lds, stream, __VA_ARGS__); \
        else RT_LAUNCH_TRACE(MODE, wide, grid, lds, stream, __VA_ARGS__);                                                 \
    } while (0)

// Global overflow part of the traversal stacks: `levels` entries for each of kOverStride lanes.  Every OWNER of
// concurrently running grids has its own buffer -- a render context (one render at a time: Context::busy), or one
// call of a stage-level test entry point -- because a lane indexes its column by its position in ITS grid only:
// two grids in flight on one device (RT_SPLIT sub-shards, host threads rendering different shard sizes) would
// otherwise push to and pop from the same columns.  Grown on demand under the owner's lock, never shrunk.
int ensure_overflow(int *&ptr, int &have_levels, int levels) {
    levels = std::max(levels, 1);
    if (ptr && have_levels >= levels) return 0;
    if (ptr) {
        HIP_TRY(hipDeviceSynchronize());  // (the owner is idle; this only guards against a caller's stray stream)
        (void)hipFree(ptr);
        ptr = nullptr;
        have_levels = 0;
    }
    HIP_TRY(hipMalloc((void **)&ptr, sizeof(int) * (size_t)levels * kOverStride));
    have_levels = levels;
    return 0;
}

int ensure_rng(Context &c, uint64_t seed, int slot_lo, hipStream_t st, double *seconds) {
    const size_t bytes = sizeof(uint32_t) * (size_t)c.n;
    uint32_t *parts[6];
    for (int k = 0; k < 6; k++) parts[k] = (uint32_t *)c.pools.array(A_RD + k);
    *seconds = 0.0;
    if (!(c.rng_valid && c.rng_seed == seed && c.rng_lo == slot_lo)) {
        HIP_TRY(hipEventRecord(c.ev_a, st));
        hipLaunchKernelGGL(k_rng_init, dim3(grid_for(c.n)), dim3(kBlock), 0, st, c.pools, c.n, slot_lo,
                           xorwow_seed(seed), c.d_jump);
        HIP_TRY(hipGetLastError());
        for (int k = 0; k < 6; k++)
            HIP_TRY(hipMemcpyAsync(c.rng_backup + (size_t)k * c.n, parts[k], bytes, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipEventRecord(c.ev_b, st));
        HIP_TRY(hipEventSynchronize(c.ev_b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c.ev_a, c.ev_b));
        *seconds = ms * 1e-3;
        c.rng_valid = true;
        c.rng_seed = seed;
        c.rng_lo = slot_lo;
    } else {
        for (int k = 0; k < 6; k++)
            HIP_TRY(hipMemcpyAsync(parts[k], c.rng_backup + (size_t)k * c.n, bytes, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

int render_shard_impl(const rt_scene *scene, const rt_camera *camera, int width, int height, int spp,
                      int max_bounces, uint64_t seed, int shard_index, int shard_count, uint32_t flags,
                      float *d_sum, hipStream_t st, rt_stats *stats, int ctx_lane = 0) {
    if (!scene || !camera || !d_sum) return fail("rt_render_shard: null argument");
    if (width <= 0 || height <= 0 || spp <= 0 || max_bounces < 0) return fail("rt_render_shard: bad dimensions");
    if (max_bounces > (1 << 24)) return fail("rt_render_shard: max_bounces exceeds 16777216");
    if (shard_count <= 0 || kW % shard_count != 0 || shard_index < 0 || shard_index >= shard_count)
        return fail("rt_render_shard: shard_count must divide 1048576 and 0 <= shard_index < shard_count");
    if ((long long)width * height > (long long)(0x7fffffff / 3))  // framebuffer values are indexed with 32 bits
        return fail("rt_render_shard: width*height exceeds 715827882 pixels");
    if (int rc = ensure_origin_radius(scene, camera->lookfrom)) return rc;  // (camera rays start at lookfrom: camera.cuh:22-26)
    // RT_FLAG_RNG_PER_SAMPLE: this rank renders the whole frame at num_samples / shard_count samples per pixel with ALL W
    // slots; camera ray `cid` of the rank has the global key cid * shard_count + shard_index, so the keys of a pixel's
    // samples are the same set whatever the shard count (see AdvanceParams)
    const bool per_sample = (flags & RT_FLAG_RNG_PER_SAMPLE) != 0;
    const bool literal = (flags & RT_FLAG_REFERENCE_WALK) != 0;
    if (literal && per_sample) return fail("rt_render_shard: RT_FLAG_REFERENCE_WALK is a parity mode and RT_FLAG_RNG_PER_SAMPLE is not: pick one");
    if (literal && (flags & RT_FLAG_WATERTIGHT)) return fail("rt_render_shard: RT_FLAG_REFERENCE_WALK and RT_FLAG_WATERTIGHT exclude each other");
    // the default: the reference's decisions (which hits its walk can see, who wins a tie) on the product's own walk.  The
    // per-sample mode is not the reference's image anyway and keeps the triangle-list definition.
    const bool verify = !literal && !per_sample && (flags & RT_FLAG_WATERTIGHT) == 0;
    if (per_sample) {
        if (spp % shard_count != 0) return fail("rt_render_shard: RT_FLAG_RNG_PER_SAMPLE needs num_samples divisible by shard_count");
        spp /= shard_count;
    }
    long long cam_end = (long long)width * height * spp;
    if (cam_end + 13LL * kW >= (1LL << 31))  // the reference's int32 camera_ray ids (render.cuh:370-371,440)
        return fail("rt_render_shard: width*height*spp exceeds the reference's int32 camera-ray range");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != scene->device) return fail("rt_render_shard: scene was created on another device");
    double ref_tree_seconds = 0.0;
    if (literal || verify) {
        const bool was_ready = scene->ref_ready;
        if (ensure_ref_tree(scene)) return 1;
        if (!was_ready) ref_tree_seconds = scene->build_seconds_ref;  // (this call paid for it)
    }
    const int n = per_sample ? kW : kW / shard_count;
    const int slot_lo = per_sample ? 0 : shard_index * n;
    Context *cp = nullptr;
    if (get_context(n, ctx_lane, &cp)) return 1;
    Context &c = *cp;
    std::lock_guard<std::mutex> busy_lock(c.busy);  // concurrent callers with the same (device, n) queue up here
    double rng_seconds = 0.0;
    if (ensure_rng(c, seed, slot_lo, st, &rng_seconds)) return 1;

    const bool time_kernels = (flags & RT_FLAG_TIME_KERNELS) != 0;
    DScene sc = scene->dev();
    Camera cam;
    memcpy(&cam, camera, sizeof(Camera));
    {
        DCounters zero{};
        zero.last_live_round = -1;
        c.h_ctr[0] = zero;  // pinned staging
        HIP_TRY(hipMemcpyAsync(c.d_ctr, &c.h_ctr[0], sizeof(DCounters), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(c.d_rows, 0, sizeof(DWaveRow) * (size_t)c.n_rows, st));  // (n / 64 rows of 64 bytes: 1 MB for the full pool)
        HIP_TRY(hipStreamSynchronize(st));  // h_ctr[0] is reused as a snapshot slot below
    }
    const int stack_cap = lds_stack_cap(scene, kLdsStack);
    const size_t lds_bytes = sizeof(int) * (size_t)kBlock * (size_t)(stack_cap + 2);  // stack (+ 1 row: push_if) + pending
    // one buffer serves the context's k_trace and k_paths grids (never in flight together); k_paths keeps fewer
    // entries in LDS, so it needs the deeper overflow
    // (the literal walk of the reference's tree -- depth <= 30 -- borrows the lane's stack: reference_walk)
    const int stack_need = std::max(scene->stack_bound, (literal || verify) ? 32 : 0);
    if (ensure_overflow(c.d_over, c.over_levels, stack_need - std::min(stack_cap, lds_stack_cap(scene, kPathsLdsStack)))) return 1;
    int *const d_over = c.d_over;
    // Slots of this shard that ever get a camera ray: slot s serves the camera rays s, s + W, ..., so in a frame of fewer than
    // W camera rays the slots from cam_end on never do anything -- the kernels of such a frame (it is nothing but the
    // lockstep rounds: 256 x 256 x 4 uses a quarter of the slots) are launched over the live slots only.  Frames of more
    // than one generation: all n.  (Per-sample streams: a lane draws its camera rays whatever its slot: all n.)
    const int n_live = per_sample ? n : (int)std::max<long long>(1, std::min<long long>(n, cam_end - (long long)slot_lo));
    hipLaunchKernelGGL(k_pool_init, dim3(grid_for(n_live)), dim3(kBlock), 0, st, c.pools, n_live, max_bounces);
    HIP_TRY(hipGetLastError());

    AdvanceParams ap;
    ap.n = n_live;
    ap.slot_lo = slot_lo;
    ap.width = width;
    ap.height = height;
    ap.spp = spp;
    ap.max_bounces = max_bounces;
    ap.cam_end = cam_end;
    ap.last_gen = per_sample ? 0x7fffffff : (int)((cam_end + kW - 1) / kW) - 1;  // (per-sample streams: no lockstep final generation)
    ap.per_sample = per_sample ? 1 : 0;
    ap.key_mul = per_sample ? shard_count : 1;
    ap.key_add = per_sample ? shard_index : 0;
    ap.seed_lo = (uint32_t)seed;
    ap.seed_hi = (uint32_t)(seed >> 32);
    ap.round = 0;
    ap.batch_mask = 7;
    ap.lockstep = 0;
    ap.fb_fixed = (flags & kFlagFixedFb) ? 1 : 0;
    ap.w_over_spp = (kW % spp == 0) ? kW / spp : 0;
    ap.dpx = ap.w_over_spp % width;
    ap.dpy = (ap.w_over_spp > 0 && width < 32768 && height < 32768) ? ap.w_over_spp / width : -1;
    const bool lds_tables = scene->n_mats <= kLdsTable && scene->n_lights <= kLdsTable;

    struct EventPair {  // destroyed on every return path
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } frame_events;
    HIP_TRY(hipEventCreate(&frame_events.a));
    HIP_TRY(hipEventCreate(&frame_events.b));
    const hipEvent_t ev_start = frame_events.a, ev_stop = frame_events.b;
    HIP_TRY(hipEventRecord(ev_start, st));

    // Rounds are enqueued in batches; after each batch the counters are snapshotted into pinned
    // host memory.  The host looks at the snapshot of batch b-2 before enqueuing batch b, so the
    // GPU always has work queued, and stops when a whole batch traced no ray.
    const int kBatch = 8;  // == ap.batch_mask + 1
    const long long generations = (cam_end + kW - 1) / kW;
    const long long max_rounds = (generations + 1) * (long long)(max_bounces + 2) + 64;
    long long rounds = 0;
    int batch = 0;
    bool finished = false;
    const dim3 grid(grid_for(n_live)), block(kBlock);
    // persistent trace kernels: as many workgroups as the chip keeps resident (never more than the
    // advance grid, whose wave count sizes the counter rows)
    int dev_cus = 0, occ_c = 0;
    HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (scene->wide)
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, k_trace<MODE_POOL, true>, kBlock, lds_bytes));
    else
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, k_trace<MODE_POOL, false>, kBlock, lds_bytes));
    int per_cu = std::max(1, occ_c);
    if (const char *e = knob("RT_TRACE_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e)));
    const int resident = std::max(1, dev_cus * per_cu);
    const dim3 grid_trace(std::min(grid_for(n_live), resident));
    TraceParams tpp{};
    tpp.total = n_live;
    tpp.fb = d_sum;
    tpp.rows = c.d_rows;
    tpp.debug_no_deposit = (flags & 0x100u) ? 1 : 0;
    tpp.fb_fixed = (flags & kFlagFixedFb) ? 1 : 0;
    tpp.vstat = &c.d_ctr->vstat[0];
#ifdef RT_TRACE_PROFILE
    unsigned long long *d_prof = nullptr;
    HIP_TRY(hipMalloc((void **)&d_prof, sizeof(unsigned long long) * 16));
    HIP_TRY(hipMemset(d_prof, 0, sizeof(unsigned long long) * 16));
    tpp.prof = d_prof;
#endif
    // RT_FLAG_TIME_KERNELS: every kTimeStride-th round is bracketed with HIP events on the launch
    // stream (no host synchronisation); the events are resolved after the loop.
    const int kTimeStride = 4;
    std::vector<hipEvent_t> &evs = c.timing_events;
    size_t ev_used = 0;
    auto next_event = [&](hipEvent_t *out) -> int {
        if (ev_used == evs.size()) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            evs.push_back(e);
        }
        *out = evs[ev_used++];
        return 0;
    };
    // ---- asynchronous part of the frame: ONE persistent launch (k_paths), or -- RT_PERSISTENT=0 -- the
    // round-per-launch pipeline (k_advance + k_trace) that the lockstep final generation also uses
    bool persistent = true;
    if (const char *e = knob("RT_PERSISTENT")) persistent = atoi(e) != 0;
    if (per_sample && !persistent) return fail("rt_render_shard: RT_FLAG_RNG_PER_SAMPLE runs on the persistent kernel only");
    float ms_paths = 0.f;
    int top_records_in_lds = 0;
    if (persistent) {
        const int paths_cap = lds_stack_cap(scene, kPathsLdsStack);
        int *const d_over2 = d_over;
        size_t lds_paths = sizeof(int) * (size_t)kBlock * (size_t)(paths_cap + 26) + (lds_tables ? sizeof(float) * (size_t)((scene->tab_dwords + 3) & ~3) : 0) +
                           sizeof(Camera) + sizeof(AdvanceParams);
        bool majority = true;
        if (const char *e = knob("RT_MAJORITY")) majority = atoi(e) != 0;
        const int dbg = (flags & 0x100u) ? 1 : 0;
        unsigned long long *paths_prof = nullptr;
#ifdef RT_TRACE_PROFILE
        const size_t prof_bytes = 192 + 32 * (size_t)(2 * grid_for(n) * (kBlock / 64));  // (x 2: RT_HALF_WAVES)
        HIP_TRY(hipMalloc((void **)&paths_prof, prof_bytes));
        HIP_TRY(hipMemset(paths_prof, 0, prof_bytes));
#endif
        // all workgroups resident at once (4 per CU at <= 128 VGPRs), lane count a divisor of n
        int paths_blocks = grid_for(n);
        // RT_HALF_WAVES=1 (experiment, shards of <= 1/8 of the slots): 32 slots per wave instead of 64, twice the waves
        int half_fill = 0;
        if (const char *e = knob("RT_HALF_WAVES")) half_fill = (atoi(e) != 0 && 2 * paths_blocks <= 1024 && !per_sample && 2 * paths_blocks * (kBlock / 64) <= c.n_rows) ? 1 : 0;  // (rows: see get_context)
        if (half_fill) paths_blocks *= 2;
        {
            int want = 1024;
            if (const char *e = knob("RT_PATHS_BLOCKS")) want = std::max(1, atoi(e));
            while (paths_blocks > want && paths_blocks % 2 == 0) paths_blocks /= 2;
        }
        const dim3 grid_paths(paths_blocks);
        int dev_cus_paths = 0;
        HIP_TRY(hipDeviceGetAttribute(&dev_cus_paths, hipDeviceAttributeMultiprocessorCount, dev));
        const bool few_blocks = paths_blocks <= 2 * dev_cus_paths;
        int top_n = 0;
        if (few_blocks) {
            // records of the top of the tree kept in LDS (within the 64 KB of dynamic LDS a launch gets without further
            // ado, ~36 KB of it slot state): 384 records of the binary tree.  For the 4-wide tree the copy buys nothing
            // (1/8 shard of C2: 2 013 / 2 017 / 2 016 / 2 015 Msamples/s with 0 / 64 / 128 / 224 nodes in LDS -- the
            // first levels are L2 hits the two waves' other work hides), so it is off unless RT_TOP_NODES asks for it
            const int prefix = std::min(scene->n_nodes, (int)rtbvh::kTopPrefix * (scene->wide ? 2 : 1));
            top_n = scene->top_prefix ? std::min(scene->wide ? 0 : 384, prefix) : 0;
            if (const char *e = knob("RT_TOP_NODES")) top_n = scene->top_prefix ? std::max(0, std::min(std::min(768, atoi(e)), prefix)) : 0;
            if (scene->wide) top_n &= ~1;  // whole nodes
            // (never more than the 64 KB of dynamic LDS a launch gets without further ado: scenes with many materials / lights
            // have larger tables)
            const size_t room = lds_paths < 65536 ? (65536 - lds_paths) / 64 : 0;
            top_n = (int)std::min<size_t>((size_t)top_n, room) & (scene->wide ? ~1 : ~0);
            lds_paths += (size_t)top_n * 64;
            top_records_in_lds = top_n;
        }
        // lanes waiting for the ADV block before it runs: full pool flat 16..24 (round 3, with the triangle block behind the
        // node block and no trip through the loop head after ADV / GEN: 20 and GEN 6 are 1 % ahead of 24 and 8); the
        // 2-waves-per-SIMD shards want 30..38 (24: -2.5 %)
        int adv_batch = few_blocks ? 34 : 20;
        int gen_batch = 6;  // lanes waiting for the GEN block before it runs (unless nothing else can); flat 4..8
        if (half_fill) {  // (half the lanes per wave: half the thresholds)
            adv_batch /= 2;
            gen_batch /= 2;
        }
        if (const char *e = knob("RT_ADV_BATCH")) adv_batch = std::max(1, std::min(64, atoi(e)));
        if (const char *e = knob("RT_GEN_BATCH")) gen_batch = std::max(1, std::min(64, atoi(e)));
        int tri_follow = 1;  // a triangle block right behind a node block when this many lanes hold a leaf by then; 0 = never
        if (const char *e = knob("RT_TRI_FOLLOW")) tri_follow = std::max(0, std::min(64, atoi(e)));
        // log2 of the priority-rotation period in scheduling decisions; 0 = off (128 ms for C2's frame).  Full pool: 111.7 - 112.2 /
        // 111.8 / 111.9 / 112.1 / 112.5 ms at 4 / 5 / 6 / 7 / 8 (late round 5; C3 -0.9 % at 4, C4 flat); 1/8 shards want 8 (+1 % at 4)
        int prio_rotate = few_blocks ? 8 : 5;
        // period, in 64-slot blocks, after which slots repeat the same pixel-column lattice (see k_paths)
        int rot_wave = 0, rot_set = 0;
        {
            long long period = 0;
            if (spp % 64 == 0 && kW % spp == 0) {
                const long long step = (kW / spp) % width;  // columns a slot moves per generation
                long long a = step, b = width;
                while (b) { long long t = a % b; a = b; b = t; }
                period = a * (spp / 64);  // gcd(step, width) columns x blocks per pixel
            }
            const int waves = paths_blocks * (kBlock / 64);
            if (period < 16 || period > waves) period = std::max(16, waves / 8);
            rot_wave = (int)(period / 4);                // measured best on the bunny scenes: 128 / 160 blocks
            rot_set = (int)(period / 4 + period / 16);
            if (const char *e = knob("RT_ROT_WAVE")) rot_wave = atoi(e);
            if (const char *e = knob("RT_ROT_SET")) rot_set = atoi(e);
            rot_wave &= ~3;  // keeps wave j of a workgroup on blocks = j (mod 4): the map stays a bijection
        }
        if (const char *e = knob("RT_PRIO_ROTATE")) prio_rotate = atoi(e);
        HIP_TRY(hipEventRecord(c.ev_a, st));
// MIN_WAVES: 4 waves per SIMD (at most 128 VGPRs) when the grid fills the chip, 2 (up to 256 VGPRs) when the
        // shard is so small that only 2 workgroups per CU exist anyway (8-GPU runs)
#define RT_LAUNCH_PATHS_V(T, WD, MJ, VER)                                                                              \
    do {                                                                                                               \
        if (few_blocks)                                                                                                \
            hipLaunchKernelGGL((k_paths<T, WD, MJ, 2, false, false, VER>), grid_paths, block, lds_paths, st, sc, c.pools, cam, ap, d_sum,   \
                               c.d_rows, paths_cap, d_over2, adv_batch, dbg, paths_prof, top_n, prio_rotate, rot_wave, rot_set, gen_batch, tri_follow, &c.d_ctr->pad2[0], half_fill, &c.d_ctr->vstat[0]); \
        else                                                                                                           \
            hipLaunchKernelGGL((k_paths<T, WD, MJ, 4, false, false, VER>), grid_paths, block, lds_paths, st, sc, c.pools, cam, ap, d_sum,   \
                               c.d_rows, paths_cap, d_over2, adv_batch, dbg, paths_prof, 0, prio_rotate, rot_wave, rot_set, gen_batch, tri_follow, &c.d_ctr->pad2[0], half_fill, &c.d_ctr->vstat[0]);     \
    } while (0)
#define RT_LAUNCH_PATHS(T, WD, MJ)                 \
    do {                                           \
        if (verify) RT_LAUNCH_PATHS_V(T, WD, MJ, true); \
        else RT_LAUNCH_PATHS_V(T, WD, MJ, false);  \
    } while (0)
        if (literal) {
            // RT_FLAG_REFERENCE_WALK: the build whose node block is the reference's own walk
            // (full pool: the 4-wave build although it spills 53 VGPRs -- measured 1 082 ms for C2's frame against 1 412 ms
            // for the spill-free 2-wave build at half the occupancy)
#define RT_LAUNCH_REF(T)                                                                                               \
    do {                                                                                                               \
        if (few_blocks)                                                                                                \
            hipLaunchKernelGGL((k_paths<T, false, true, 2, false, true>), grid_paths, block, lds_paths, st, sc, c.pools, cam, ap, d_sum, \
                               c.d_rows, paths_cap, d_over2, adv_batch, dbg, paths_prof, 0, prio_rotate, rot_wave, rot_set, gen_batch, tri_follow, &c.d_ctr->pad2[0], half_fill, &c.d_ctr->vstat[0]); \
        else                                                                                                           \
            hipLaunchKernelGGL((k_paths<T, false, true, 4, false, true>), grid_paths, block, lds_paths, st, sc, c.pools, cam, ap, d_sum, \
                               c.d_rows, paths_cap, d_over2, adv_batch, dbg, paths_prof, 0, prio_rotate, rot_wave, rot_set, gen_batch, tri_follow, &c.d_ctr->pad2[0], half_fill, &c.d_ctr->vstat[0]); \
    } while (0)
            if (lds_tables) RT_LAUNCH_REF(true);
            else RT_LAUNCH_REF(false);
#undef RT_LAUNCH_REF
        } else if (per_sample && !few_blocks) {
            // per-sample streams: the build in which the waves draw their camera rays from the frame's counter
#define RT_LAUNCH_DRAW(T, WD)                                                                                          \
    hipLaunchKernelGGL((k_paths<T, WD, true, 4, true>), grid_paths, block, lds_paths, st, sc, c.pools, cam, ap, d_sum,  \
                       c.d_rows, paths_cap, d_over2, adv_batch, dbg, paths_prof, 0, prio_rotate, rot_wave, rot_set, gen_batch, tri_follow, &c.d_ctr->pad2[0], half_fill, &c.d_ctr->vstat[0])
            if (lds_tables && scene->wide) RT_LAUNCH_DRAW(true, true);
            else if (lds_tables) RT_LAUNCH_DRAW(true, false);
            else if (scene->wide) RT_LAUNCH_DRAW(false, true);
            else RT_LAUNCH_DRAW(false, false);
#undef RT_LAUNCH_DRAW
        } else if (majority) {
            if (lds_tables && scene->wide) RT_LAUNCH_PATHS(true, true, true);
            else if (lds_tables) RT_LAUNCH_PATHS(true, false, true);
            else if (scene->wide) RT_LAUNCH_PATHS(false, true, true);
            else RT_LAUNCH_PATHS(false, false, true);
        } else {
            if (lds_tables && scene->wide) RT_LAUNCH_PATHS(true, true, false);
            else if (lds_tables) RT_LAUNCH_PATHS(true, false, false);
            else if (scene->wide) RT_LAUNCH_PATHS(false, true, false);
            else RT_LAUNCH_PATHS(false, false, false);
        }
#undef RT_LAUNCH_PATHS
#undef RT_LAUNCH_PATHS_V
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c.ev_b, st));
        HIP_TRY(hipEventSynchronize(c.ev_b));
        HIP_TRY(hipEventElapsedTime(&ms_paths, c.ev_a, c.ev_b));
#ifdef RT_TRACE_PROFILE
        {
            unsigned long long h[24];
            HIP_TRY(hipMemcpy(h, paths_prof, 192, hipMemcpyDeviceToHost));
            fprintf(stderr, "k_paths cycles: ADV %.1f%% (%.0f / block) node %.1f%% (%.0f) tri %.1f%% (%.0f) rest %.1f%%\n",
                    100.0 * h[8] / h[11], h[0] ? (double)h[8] / h[0] : 0.0, 100.0 * h[9] / h[11], h[2] ? (double)h[9] / h[2] : 0.0,
                    100.0 * h[10] / h[11], h[4] ? (double)h[10] / h[4] : 0.0, 100.0 * (double)(h[11] - h[8] - h[9] - h[10]) / h[11]);
            fprintf(stderr, "k_paths GEN blocks: %llu avg lanes %.1f, %.1f %% of wave time (%.0f cycles / block; inside `rest` above)\n", h[12],
                    h[12] ? (double)h[15] / h[12] : 0.0, 100.0 * h[16] / h[11], h[12] ? (double)h[16] / h[12] : 0.0);
            fprintf(stderr, "k_paths waves: %llu, mean lifetime %.0f cycles, longest %.0f cycles (x%.3f)\n", h[14], (double)h[11] / h[14], (double)h[13],
                    (double)h[13] * h[14] / h[11]);
            fprintf(stderr, "k_paths fin section: %.1f %% of wave time, entered in %llu iterations (avg %.1f finished lanes), %.0f cycles each\n",
                    100.0 * h[17] / h[11], h[19], h[19] ? (double)h[18] / h[19] : 0.0, h[19] ? (double)h[17] / h[19] : 0.0);
            fprintf(stderr, "k_paths profile: ADV blocks %llu avg lanes %.1f | node steps %llu avg lanes %.1f (ADV-waiting %.1f) | tri steps %llu avg lanes %.1f (ADV-waiting %.1f)\n",
                    h[0], h[0] ? (double)h[1] / h[0] : 0.0, h[2], h[2] ? (double)h[3] / h[2] : 0.0, h[2] ? (double)h[6] / h[2] : 0.0, h[4],
                    h[4] ? (double)h[5] / h[4] : 0.0, h[4] ? (double)h[7] / h[4] : 0.0);
            if (const char *dump = knob("RT_PROF_DUMP")) {  // per-wave records: hw_id xcc_id cycles blocks
                std::vector<unsigned long long> recs(4 * (size_t)paths_blocks * (kBlock / 64));
                HIP_TRY(hipMemcpy(recs.data(), paths_prof + 24, recs.size() * 8, hipMemcpyDeviceToHost));
                if (FILE *f = fopen(dump, "w")) {
                    for (size_t k = 0; k < recs.size(); k += 4)
                        fprintf(f, "%zu %llu %llu %llu %llu\n", k / 4, recs[k], recs[k + 1], recs[k + 2], recs[k + 3]);
                    fclose(f);
                }
            }
            (void)hipFree(paths_prof);
        }
#endif
        finished = true;
    }
    while (!finished && rounds < max_rounds) {
        for (int k = 0; k < kBatch; k++) {
            ap.round = (int)(rounds & 0x3fffffff);
            if (time_kernels && (rounds % kTimeStride) == 0) {
                hipEvent_t e0, e1, e2, e3;
                if (next_event(&e0) || next_event(&e1) || next_event(&e2) || next_event(&e3)) return 1;
                HIP_TRY(hipEventRecord(e0, st));
                RT_LAUNCH_ADVANCE(st, d_sum);
                HIP_TRY(hipEventRecord(e1, st));
                RT_LAUNCH_TRACE_REF(MODE_POOL, literal, verify, scene->wide, grid_trace, lds_bytes, st, sc, c.pools, tpp, stack_cap, d_over);
                HIP_TRY(hipEventRecord(e2, st));
                HIP_TRY(hipEventRecord(e3, st));
            } else {
                RT_LAUNCH_ADVANCE(st, d_sum);
                RT_LAUNCH_TRACE_REF(MODE_POOL, literal, verify, scene->wide, grid_trace, lds_bytes, st, sc, c.pools, tpp, stack_cap, d_over);
            }
            rounds++;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&c.h_ctr[batch & 3], c.d_ctr, sizeof(DCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(c.ev_ring[batch & 3], st));
        if (batch >= 1) {
            int prev = (batch - 1) & 3;
            HIP_TRY(hipEventSynchronize(c.ev_ring[prev]));
            // k_advance records liveness only in the round that closes a batch (liveness is monotone):
            // if the closing round of batch b-1 traced nothing, every slot is finished
            if ((long long)c.h_ctr[prev].last_live_round < (long long)batch * kBatch - 1) finished = true;
        }
        batch++;
    }
    // ---- final generation in lockstep (see k_advance): round 0 generates, every later round is one
    // reference iteration; the render ends at the first round in which nothing shades (render.cuh:436)
    // All max_bounces + 2 rounds are enqueued back to back; which of them still belong to the render is decided on the
    // device (k_advance / k_trace look at lock_shades), and the host learns the number of rounds that ran afterwards.
    // The rounds go out in chunks of kLockChunk: the reference's 12 rounds (max_bounces = 10) are ONE chunk -- no host
    // read-back inside the frame --, and a caller with max_bounces in the thousands does not pay thousands of empty launches
    // once the render has ended: between chunks the host looks at the last round's counter.
    const int n_lock = max_bounces + 2;
    const bool ran_lockstep = finished && !per_sample;
    int lock_enqueued = 0;
    if (ran_lockstep) {
        if (n_lock > c.lock_cap) {  // (the stop rule's counters: one word per round, grown on demand)
            HIP_TRY(hipStreamSynchronize(st));
            (void)hipFree(c.d_lock);
            c.d_lock = nullptr;
            c.lock_cap = 0;
            HIP_TRY(hipMalloc((void **)&c.d_lock, sizeof(unsigned) * (size_t)std::max(n_lock, 64)));
            c.lock_cap = std::max(n_lock, 64);
        }
        HIP_TRY(hipMemsetAsync(c.d_lock, 0, sizeof(unsigned) * (size_t)n_lock, st));
        while (lock_enqueued < n_lock) {
            const int hi = std::min(n_lock, lock_enqueued + kLockChunk);
            for (int j = lock_enqueued; j < hi; j++) {
                ap.round = (int)((rounds + j) & 0x3fffffff);
                ap.lockstep = 1 + j;
                RT_LAUNCH_ADVANCE(st, d_sum);
                tpp.lock_shades = c.d_lock;
                tpp.lock_round = j;
                RT_LAUNCH_TRACE_REF(MODE_POOL, literal, verify, scene->wide, grid_trace, lds_bytes, st, sc, c.pools, tpp, stack_cap, d_over);
            }
            HIP_TRY(hipGetLastError());
            lock_enqueued = hi;
            if (hi < n_lock) {  // (more than one chunk: max_bounces >= 15)
                unsigned last = 1u;
                HIP_TRY(hipMemcpyAsync(&last, c.d_lock + (hi - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (last == 0u && hi - 1 >= 1) break;  // that round shaded nothing: the render is over (render.cuh:436)
            }
        }
    }
    HIP_TRY(hipEventRecord(ev_stop, st));
    HIP_TRY(hipEventSynchronize(ev_stop));
    if (ran_lockstep) {  // rounds that ran: up to and including the first one (after the generating round) that shaded nothing
        std::vector<unsigned> h_lock((size_t)lock_enqueued);
        HIP_TRY(hipMemcpy(h_lock.data(), c.d_lock, sizeof(unsigned) * (size_t)lock_enqueued, hipMemcpyDeviceToHost));
        int ran = lock_enqueued;
        for (int j = 1; j < lock_enqueued; j++)
            if (h_lock[(size_t)j] == 0u) {
                ran = j + 1;
                break;
            }
        rounds += ran;
    }
#ifdef RT_TRACE_PROFILE
    {
        unsigned long long h[16];
        HIP_TRY(hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "trace profile: waves %llu outer_it %llu refills %llu | inner_it %llu avg_lanes %.1f | leaf_it %llu avg_lanes %.1f | "
                        "tri_it %llu avg_lanes %.1f | active lanes at loop top %.1f | finalised lanes per refill %.1f\n",
                h[10], h[0], h[1], h[2], h[2] ? (double)h[3] / h[2] : 0.0, h[4], h[4] ? (double)h[5] / h[4] : 0.0, h[6],
                h[6] ? (double)h[7] / h[6] : 0.0, h[0] ? (double)h[8] / h[0] : 0.0, h[1] ? (double)h[9] / h[1] : 0.0);
        (void)hipFree(d_prof);
    }
#endif
    float ms_total = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms_total, ev_start, ev_stop));
    DCounters h_final{};
    HIP_TRY(hipMemcpy(&h_final, c.d_ctr, sizeof(DCounters), hipMemcpyDeviceToHost));
    std::vector<DWaveRow> h_rows((size_t)c.n_rows);
    HIP_TRY(hipMemcpy(h_rows.data(), c.d_rows, sizeof(DWaveRow) * (size_t)c.n_rows, hipMemcpyDeviceToHost));
    unsigned long long fin[C_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const DWaveRow &r : h_rows)
        for (int k = 0; k < C_COUNT; k++) fin[k] += r.c[k];
    double t_adv = 0, t_ch = 0, t_ah = 0;
    long long n_sampled = (long long)(ev_used / 4);
    for (size_t q = 0; q + 3 < ev_used; q += 4) {
        float ms;
        HIP_TRY(hipEventElapsedTime(&ms, evs[q], evs[q + 1]));
        t_adv += ms;
        HIP_TRY(hipEventElapsedTime(&ms, evs[q + 1], evs[q + 2]));
        t_ch += ms;
        HIP_TRY(hipEventElapsedTime(&ms, evs[q + 2], evs[q + 3]));
        t_ah += ms;
    }
    if (!finished) return fail("rt_render_shard: round limit reached before the path pool drained");
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->camera_rays = (int64_t)fin[C_CAMERA];
        stats->shade_events = (int64_t)fin[C_SHADE];
        stats->closest_rays = (int64_t)fin[C_CLOSEST];
        stats->any_rays = (int64_t)fin[C_ANY];
        stats->emission_adds = (int64_t)fin[C_EMIT];
        stats->shadow_adds = (int64_t)fin[C_SHADOW_ADD];
        stats->rr_draws = (int64_t)fin[C_RR];
        stats->iterations = rounds;
        stats->bvh_nodes = scene->n_nodes;
        stats->bvh_depth = scene->max_depth;
        stats->seconds_render = ms_total * 1e-3;
        stats->seconds_rng_init = rng_seconds;
        // sampled every kTimeStride-th round; scaled to all rounds (average launch duration x launches)
        double scale_up = n_sampled > 0 ? (double)rounds / (double)n_sampled : 0.0;
        stats->seconds_trace = t_ch * 1e-3 * scale_up;
        stats->seconds_reference_tree = ref_tree_seconds;
        (void)t_ah;
        stats->seconds_advance = t_adv * 1e-3 * scale_up;
        stats->launches_trace = rounds;
        stats->reserved[0] = n_sampled;
        if (persistent) {  // the frame's dominant kernel is the one k_paths launch
            stats->seconds_trace = ms_paths * 1e-3;
            stats->seconds_advance = 0.0;
            stats->launches_trace = 1;
            stats->reserved[0] = 1;
            stats->reserved[1] = 1;
            stats->reserved[2] = top_records_in_lds;
        }
        stats->reserved[4] = (int64_t)h_final.vstat[V_LITERAL];
        stats->reserved[5] = (int64_t)h_final.vstat[V_LOST];
        stats->reserved[6] = (int64_t)h_final.vstat[V_TIE];
    }
    return 0;
}

// A shard may be rendered as `split` interleaved sub-shards on separate HIP streams (one host thread
// each): slot sets are independent, so the sub-shards only meet in the framebuffer atomics, and the
// tail of one sub-shard's persistent trace kernel overlaps the head of the other's.
int render_overlapped(const rt_scene *scene, const rt_camera *camera, int width, int height, int spp,
                      int max_bounces, uint64_t seed, int shard_index, int shard_count, uint32_t flags,
                      float *d_sum, hipStream_t st, rt_stats *stats) {
    int split = 1;
    if (const char *e = knob("RT_SPLIT")) split = std::max(1, std::min(8, atoi(e)));
    while (split > 1 && (shard_count <= 0 || kW % (shard_count * split) != 0 || kW / (shard_count * split) < 4096)) split >>= 1;
    if (split <= 1)
        return render_shard_impl(scene, camera, width, height, spp, max_bounces, seed, shard_index, shard_count, flags,
                                 d_sum, st, stats);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipStreamSynchronize(st));  // the caller zeroed d_sum on its stream
    std::vector<rt_stats> sub(split);
    std::vector<int> rc(split, 0);
    std::vector<std::string> err(split);
    std::vector<std::thread> th;
    for (int k = 0; k < split; k++)
        th.emplace_back([&, k] {
            if (hipSetDevice(dev) != hipSuccess) { rc[k] = 1; err[k] = "hipSetDevice failed"; return; }
            hipStream_t s2;
            if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { rc[k] = 1; err[k] = "stream create failed"; return; }
            rc[k] = render_shard_impl(scene, camera, width, height, spp, max_bounces, seed, shard_index * split + k,
                                      shard_count * split, flags, d_sum, s2, &sub[k], k + 1);
            if (rc[k]) err[k] = g_last_error;
            (void)hipStreamSynchronize(s2);
            (void)hipStreamDestroy(s2);
        });
    for (auto &t : th) t.join();
    for (int k = 0; k < split; k++)
        if (rc[k]) return fail(err[k]);
    if (stats) {
        rt_stats tot = sub[0];
        for (int k = 1; k < split; k++) {
            tot.camera_rays += sub[k].camera_rays;
            tot.shade_events += sub[k].shade_events;
            tot.closest_rays += sub[k].closest_rays;
            tot.any_rays += sub[k].any_rays;
            tot.emission_adds += sub[k].emission_adds;
            tot.shadow_adds += sub[k].shadow_adds;
            tot.rr_draws += sub[k].rr_draws;
            tot.iterations += sub[k].iterations;
            tot.launches_trace += sub[k].launches_trace;
            tot.seconds_trace += sub[k].seconds_trace;
            tot.seconds_advance += sub[k].seconds_advance;
            tot.seconds_render = std::max(tot.seconds_render, sub[k].seconds_render);
            tot.seconds_rng_init = std::max(tot.seconds_rng_init, sub[k].seconds_rng_init);
            tot.seconds_reference_tree = std::max(tot.seconds_reference_tree, sub[k].seconds_reference_tree);
            tot.reserved[0] += sub[k].reserved[0];
            tot.reserved[2] = std::max(tot.reserved[2], sub[k].reserved[2]);
            for (int q = 4; q < 7; q++) tot.reserved[q] += sub[k].reserved[q];
        }
        *stats = tot;
    }
    return 0;
}

}  // namespace

// ============================================================================ C-ABI
extern "C" {

const char *rt_last_error(void) { return g_last_error.c_str(); }
const char *rt_peer_access_log(void) { return g_peer_log.c_str(); }
const char *rt_version(void) { return "rtcuda_amd 0.1 (gfx950)"; }
#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char *rt_build_id(void) { return RT_BUILD_ID; }

int rt_scene_create(const float *tri_p0p1p2, int n_tris, const int32_t *tri_material, const int32_t *tri_light,
                    const rt_material *materials, int n_materials, const rt_light *lights, int n_lights,
                    rt_scene **out_scene) {
    if (!out_scene) return fail("rt_scene_create: out_scene is null");
    *out_scene = nullptr;
    if (n_tris < 0 || n_materials < 0 || n_lights < 0) return fail("rt_scene_create: negative count");
    if (n_tris >= (1 << 24)) return fail("rt_scene_create: more than 2^24 - 1 triangles (24-bit triangle addressing)");
    if (n_tris > 0 && (!tri_p0p1p2 || !tri_material)) return fail("rt_scene_create: null triangle arrays");
    if (n_tris > 0 && (n_materials == 0 || !materials)) return fail("rt_scene_create: no materials");
    if (n_lights > 0 && !lights) return fail("rt_scene_create: null lights");
    if (n_materials > 65535 || n_lights > 32766) return fail("rt_scene_create: at most 65535 materials and 32766 lights");
    for (int i = 0; i < n_tris; i++) {
        if (tri_material[i] < 0 || tri_material[i] >= n_materials)
            return fail("rt_scene_create: tri_material[" + std::to_string(i) + "] out of range");
        if (tri_light && (tri_light[i] < -1 || tri_light[i] >= n_lights))
            return fail("rt_scene_create: tri_light[" + std::to_string(i) + "] out of range");
    }
    for (int i = 0; i < n_materials; i++)
        if (materials[i].type < RT_MATTE || materials[i].type > RT_GLASS)
            return fail("rt_scene_create: unknown material type");
    for (int i = 0; i < n_lights; i++) {
        if (lights[i].type != RT_POINT_LIGHT && lights[i].type != RT_AREA_LIGHT)
            return fail("rt_scene_create: unknown light type");
        if (lights[i].type == RT_AREA_LIGHT && (lights[i].triangle < 0 || lights[i].triangle >= n_tris))
            return fail("rt_scene_create: area light triangle out of range");
    }
    auto sc = std::make_unique<rt_scene>();
    HIP_TRY(hipGetDevice(&sc->device));
    bool use_lbvh = false;
    if (const char *e = knob("RT_BVH_BUILDER")) use_lbvh = std::string(e) == "lbvh";
    rtbvh::Result bvh;
    if (use_lbvh && n_tris >= 2) {
        int depth = 0;
        if (build_lbvh_device(tri_p0p1p2, n_tris, bvh.pairs, bvh.order, depth, sc->build_seconds)) return 1;
        if (!validate_pairs(bvh.pairs, n_tris) || depth < 1) return fail("rt_scene_create: device BVH build produced a malformed tree");
        std::vector<char> seen((size_t)n_tris, 0);
        for (int k = 0; k < n_tris; k++) {
            if (bvh.order[k] < 0 || bvh.order[k] >= n_tris || seen[bvh.order[k]]) return fail("rt_scene_create: device BVH build produced a bad triangle order");
            seen[bvh.order[k]] = 1;
        }
        bvh.pair_depth = depth;
        rtbvh::quads_from_pairs(bvh);  // (host, a few ms: the 4-wide format of the same tree)
        bvh.num_leaves = n_tris;
        sc->builder = 1;
        sc->top_prefix = false;
    } else {
        auto t0 = std::chrono::steady_clock::now();
        bvh = rtbvh::build(tri_p0p1p2, n_tris);
        sc->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (!bvh.ok) return fail("rt_scene_create: BVH build produced an unreferenceable leaf");
    if (n_tris > 0 && !bvh.quads.empty() && !validate_quads(bvh.quads, n_tris))
        return fail("rt_scene_create: the 4-wide BVH is malformed (structure, or an absent child without its +inf box)");
    sc->n_tris = n_tris;
    if (n_tris > 0) sc->h_tri9.assign(tri_p0p1p2, tri_p0p1p2 + 9 * (size_t)n_tris);  // (RT_FLAG_REFERENCE_WALK builds its tree from these)
    if (n_tris > 0) sc->h_tri_material.assign(tri_material, tri_material + n_tris);
    if (n_tris > 0 && tri_light) sc->h_tri_light.assign(tri_light, tri_light + n_tris);
    if (n_materials > 0) sc->h_materials.assign(materials, materials + n_materials);
    if (n_lights > 0) sc->h_lights.assign(lights, lights + n_lights);
    sc->wide = true;  // 4-wide nodes (two pair-style records each): half the dependent fetches per ray; RT_BVH_WIDE=0: 2-wide
    if (const char *e = knob("RT_BVH_WIDE")) sc->wide = atoi(e) != 0;
    // a tree too deep for the 4-wide walk's stack (up to 3 entries per level) may still fit the 2-wide walk's (1 per level):
    // a very deep LBVH, or a host tree the reinsertion pass deepened
    if (sc->wide && bvh.stack_bound > kMaxStackBound) sc->wide = false;
    if ((sc->wide ? bvh.stack_bound : bvh.pair_depth + 1) > kMaxStackBound)
        return fail("rt_scene_create: BVH depth " + std::to_string(sc->wide ? bvh.max_depth : bvh.pair_depth) + " exceeds the traversal stack");
    sc->n_nodes = sc->wide ? (int)bvh.quads.size() : (int)bvh.pairs.size();  // 64-byte records
    sc->max_depth = sc->wide ? bvh.max_depth : bvh.pair_depth;
    sc->stack_bound = sc->wide ? bvh.stack_bound : bvh.pair_depth + 1;
    sc->n_leaves = bvh.num_leaves;
    sc->n_lights = n_lights;
    sc->n_mats = n_materials;
    sc->h_order.assign(bvh.order.begin(), bvh.order.end());
    sc->h_inverse.assign(n_tris, 0);
    for (int k = 0; k < n_tris; k++) sc->h_inverse[sc->h_order[k]] = k;
    // triangle records in leaf order: e1 = p0 - p1, e2 = p2 - p0, n = e1 x e2 (triangle.cuh:6-7),
    // computed here in fp32 without contraction (this file is built with -ffp-contract=off)
    std::vector<float> trec((size_t)12 * std::max(n_tris, 1));
    std::vector<int2> info(std::max(n_tris, 1));
    for (int k = 0; k < n_tris; k++) {
        int i = sc->h_order[k];
        const float *q = tri_p0p1p2 + 9 * (size_t)i;
        float p0[3] = {q[0], q[1], q[2]};
        float e1[3] = {q[0] - q[3], q[1] - q[4], q[2] - q[5]};
        float e2[3] = {q[6] - q[0], q[7] - q[1], q[8] - q[2]};
        float nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        float *r = trec.data() + 12 * (size_t)k;
        r[0] = p0[0]; r[1] = p0[1]; r[2] = p0[2];
        r[3] = e1[0]; r[4] = e1[1]; r[5] = e1[2];
        r[6] = e2[0]; r[7] = e2[1]; r[8] = e2[2];
        r[9] = nn[0]; r[10] = nn[1]; r[11] = nn[2];
        info[k] = make_int2(tri_material[i], tri_light ? tri_light[i] : -1);
    }
    std::vector<Light> dl(std::max(n_lights, 1));
    for (int i = 0; i < n_lights; i++) {
        memcpy(&dl[i], &lights[i], sizeof(Light));
        if (lights[i].type == RT_AREA_LIGHT) dl[i].tri = sc->h_inverse[lights[i].triangle];
    }
    static_assert(sizeof(Light) == sizeof(rt_light), "light layout");
    static_assert(sizeof(Material) == sizeof(rt_material), "material layout");
    static_assert(sizeof(Camera) == sizeof(rt_camera), "camera layout");
    HIP_TRY(hipMalloc((void **)&sc->d_nodes, 64 * (size_t)sc->n_nodes));
    if (sc->wide) {
        sc->h_quads = bvh.quads;
        float radius[3];
        rtbvh::quads_abs_bounds(sc->h_quads, radius);
        if (int rc = upload_node_records(sc.get(), sc->h_quads, radius)) return rc;
    } else {
        const float none[3] = {0.f, 0.f, 0.f};
        if (int rc = upload_node_records(sc.get(), bvh.pairs, none)) return rc;
    }
    HIP_TRY(hipMalloc((void **)&sc->d_tris, sizeof(float) * trec.size()));
    HIP_TRY(hipMemcpy(sc->d_tris, trec.data(), sizeof(float) * trec.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&sc->d_tri_info, sizeof(int2) * info.size()));
    HIP_TRY(hipMemcpy(sc->d_tri_info, info.data(), sizeof(int2) * info.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&sc->d_tri_shade, sizeof(float4) * std::max<size_t>(info.size(), 1)));
    if (n_tris > 0) {
        hipLaunchKernelGGL(k_build_tri_shade, dim3((n_tris + 255) / 256), dim3(256), 0, nullptr, sc->d_tris, sc->d_tri_info, n_tris,
                           sc->d_tri_shade);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMalloc((void **)&sc->d_mats, sizeof(Material) * std::max(n_materials, 1)));
    if (n_materials)
        HIP_TRY(hipMemcpy(sc->d_mats, materials, sizeof(Material) * n_materials, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&sc->d_lights, sizeof(Light) * dl.size()));
    HIP_TRY(hipMemcpy(sc->d_lights, dl.data(), sizeof(Light) * dl.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&sc->d_order, sizeof(int) * std::max(n_tris, 1)));
    if (n_tris) HIP_TRY(hipMemcpy(sc->d_order, sc->h_order.data(), sizeof(int) * n_tris, hipMemcpyHostToDevice));
    sc->tab_dwords = 5 * n_materials + 24 * n_lights;
    HIP_TRY(hipMalloc((void **)&sc->d_tables, sizeof(float) * (size_t)std::max(sc->tab_dwords, 1)));
    {
        int nt = std::max(std::max(n_materials, n_lights), 1);
        hipLaunchKernelGGL(k_build_tables, dim3((nt + 63) / 64), dim3(64), 0, nullptr, sc->d_mats, n_materials,
                           sc->d_lights, n_lights, sc->d_tris, sc->d_tables);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    *out_scene = sc.release();
    return 0;
}

void rt_scene_destroy(rt_scene *scene) { delete scene; }  // (~rt_scene frees the device arrays)

int rt_scene_info(const rt_scene *scene, int64_t out[4]) {
    if (!scene || !out) return fail("rt_scene_info: null argument");
    out[0] = scene->n_nodes;
    out[1] = scene->n_tris;
    out[2] = scene->max_depth;
    out[3] = scene->n_leaves;
    return 0;
}

int rt_scene_build_info(const rt_scene *scene, int *builder, double *seconds) {
    if (!scene || !builder || !seconds) return fail("rt_scene_build_info: null argument");
    *builder = scene->builder;
    *seconds = scene->build_seconds;
    return 0;
}

int rt_camera_make(const float lookfrom[3], const float lookat[3], const float up[3], float vfov_deg,
                   float aspect_ratio, rt_camera *out) {
    if (!lookfrom || !lookat || !up || !out) return fail("rt_camera_make: null argument");
    // camera.cuh:15-29, host fp32 (tanf from the host libm, as in the reference)
    const float pi = 3.14159265358979323846f;
    float vfov_rad = vfov_deg * (pi / 180.f);
    float vh = 2.f * tanf(vfov_rad * 0.5f);
    float vw = vh * aspect_ratio;
    auto sub3 = [](const float *a, const float *b, float *r) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; };
    auto dot3 = [](const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto unit3 = [&](float *a) {
        float inv = 1.f / sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        a[0] *= inv; a[1] *= inv; a[2] *= inv;
    };
    float w[3], v[3], u[3];
    sub3(lookfrom, lookat, w);
    unit3(w);
    float duw = dot3(up, w);
    v[0] = up[0] - duw * w[0];
    v[1] = up[1] - duw * w[1];
    v[2] = up[2] - duw * w[2];
    unit3(v);
    u[0] = v[1] * w[2] - v[2] * w[1];
    u[1] = v[2] * w[0] - v[0] * w[2];
    u[2] = v[0] * w[1] - v[1] * w[0];
    for (int a = 0; a < 3; a++) {
        out->lookfrom[a] = lookfrom[a];
        out->horizontal[a] = vw * u[a];
        out->vertical[a] = -vh * v[a];
    }
    for (int a = 0; a < 3; a++)
        out->upper_left[a] = ((lookfrom[a] - w[a]) - 0.5f * out->horizontal[a]) - 0.5f * out->vertical[a];
    return 0;
}

int rt_render_shard(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
                    int max_bounces, uint64_t seed, int shard_index, int shard_count, uint32_t flags,
                    float *d_sum_rgb, void *stream, rt_stats *stats) {
    return render_overlapped(scene, camera, width, height, num_samples, max_bounces, seed, shard_index, shard_count,
                             flags & ~kFlagFixedFb, d_sum_rgb, (hipStream_t)stream, stats);
}

int rt_render_shard_fixed(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
                          int max_bounces, uint64_t seed, int shard_index, int shard_count, uint32_t flags,
                          int64_t *d_sum_fixed, void *stream, rt_stats *stats) {
    return render_overlapped(scene, camera, width, height, num_samples, max_bounces, seed, shard_index, shard_count,
                             (flags & ~kFlagFixedFb) | kFlagFixedFb, (float *)d_sum_fixed, (hipStream_t)stream, stats);
}

int rt_post_process_fixed(const int64_t *d_sum_fixed, float *d_rgb_out, int num_pixels, int num_samples, void *stream) {
    if (!d_sum_fixed || !d_rgb_out || num_pixels <= 0 || num_samples <= 0) return fail("rt_post_process_fixed: bad argument");
    if (num_pixels > 0x7fffffff / 3) return fail("rt_post_process_fixed: more than 715827882 pixels");
    int nv = num_pixels * 3;
    float inv = 1.f / (float)num_samples;
    hipLaunchKernelGGL(k_post_process_fixed, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)d_sum_fixed, d_rgb_out, nv, inv);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rt_post_process(float *d_rgb, int num_pixels, int num_samples, void *stream) {
    if (!d_rgb || num_pixels <= 0 || num_samples <= 0) return fail("rt_post_process: bad argument");
    if (num_pixels > 0x7fffffff / 3) return fail("rt_post_process: more than 715827882 pixels");
    int nv = num_pixels * 3;
    float inv = 1.f / (float)num_samples;
    hipLaunchKernelGGL(k_post_process, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_rgb, nv, inv);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rt_render(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
              int max_bounces, uint64_t seed, uint32_t flags, float *out_rgb, rt_stats *stats) {
    if (!out_rgb) return fail("rt_render: out_rgb is null");
    if (width <= 0 || height <= 0) return fail("rt_render: bad dimensions");
    if ((long long)width * height > (long long)(0x7fffffff / 3)) return fail("rt_render: width*height exceeds 715827882 pixels");
    const bool fixed = (flags & RT_FLAG_DETERMINISTIC) != 0;
    const size_t n_values = 3 * (size_t)width * height;
    const size_t bytes = sizeof(float) * n_values;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail("rt_render: device ordinal out of range");
    std::lock_guard<std::mutex> out_lock(g_dev_busy[dev]);  // (this device's cached buffers serve one host-output call at a time)
    float *d_fb = (float *)out_buffer(dev, 0, bytes);
    long long *d_fixed = fixed ? (long long *)out_buffer(dev, 1, sizeof(long long) * n_values) : nullptr;
    if (!d_fb || (fixed && !d_fixed)) return fail("rt_render: out of device memory");
    int rc = 0;
    if (fixed) {
        HIP_TRY(hipMemsetAsync(d_fixed, 0, sizeof(long long) * n_values, nullptr));
        rc = render_overlapped(scene, camera, width, height, num_samples, max_bounces, seed, 0, 1,
                               (flags & ~kFlagFixedFb) | kFlagFixedFb, (float *)d_fixed, nullptr, stats);
        if (rc) return rc;
        rc = rt_post_process_fixed((const int64_t *)d_fixed, d_fb, width * height, num_samples, nullptr);
    } else {
        HIP_TRY(hipMemsetAsync(d_fb, 0, bytes, nullptr));
        rc = render_overlapped(scene, camera, width, height, num_samples, max_bounces, seed, 0, 1, flags & ~kFlagFixedFb,
                               d_fb, nullptr, stats);
        if (rc) return rc;
        rc = rt_post_process(d_fb, width * height, num_samples, nullptr);
    }
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_rgb, d_fb, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// Releases every device allocation the library holds behind the scenes -- the per-device render contexts (path pools, RNG
// states, counters, overflow stacks, events) and the cached output buffers of rt_render / rt_render_multi.  Scenes are the
// caller's (rt_scene_destroy).  No render may be in flight.  The library works again afterwards (everything is re-created on
// demand).  (The reference frees nothing at all: render.cuh:374-391, bvh.cuh:211-217.)
void rt_shutdown(void) {
    int saved = 0;
    const bool have = hipGetDevice(&saved) == hipSuccess;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_contexts.clear();  // (~Context frees on the context's own device)
    }
    {
        std::lock_guard<std::mutex> list_lock(g_out_mutex);
        for (OutBuffer &b : g_out_buffers) {
            if (hipSetDevice(b.device) == hipSuccess) (void)hipFree(b.ptr);
        }
        g_out_buffers.clear();
    }
    if (have) (void)hipSetDevice(saved);
}

// The scene as it exists on `device`: the scene itself, or a replica created there from the host copies (once per device).
static const rt_scene *scene_on_device(const rt_scene *scene, int device) {
    if (scene->device == device) return scene;
    std::lock_guard<std::mutex> lock(scene->replica_mutex);
    for (const rt_scene *r : scene->replicas)
        if (r->device == device) return r;
    int saved = 0;
    if (hipGetDevice(&saved) != hipSuccess || hipSetDevice(device) != hipSuccess) {
        fail("rt_render_multi: cannot select device " + std::to_string(device));
        return nullptr;
    }
    rt_scene *rep = nullptr;
    const int rc = rt_scene_create(scene->h_tri9.data(), scene->n_tris, scene->h_tri_material.data(),
                                   scene->h_tri_light.empty() ? nullptr : scene->h_tri_light.data(), scene->h_materials.data(),
                                   scene->n_mats, scene->h_lights.data(), scene->n_lights, &rep);
    (void)hipSetDevice(saved);
    if (rc != 0) return nullptr;
    scene->replicas.push_back(rep);
    return rep;
}

int rt_render_multi(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
                    int max_bounces, uint64_t seed, uint32_t flags, const int *devices, int n_devices, float *out_rgb,
                    rt_stats *stats) {
    if (!scene || !camera || !out_rgb) return fail("rt_render_multi: null argument");
    if (!devices || n_devices < 1) return fail("rt_render_multi: empty device list");
    if (width <= 0 || height <= 0) return fail("rt_render_multi: bad dimensions");
    if ((long long)width * height > (long long)(0x7fffffff / 3)) return fail("rt_render_multi: width*height exceeds 715827882 pixels");
    if (kW % n_devices != 0) return fail("rt_render_multi: the number of devices must divide 1048576 (1, 2, 4, 8, ...)");
    int n_visible = 0;
    HIP_TRY(hipGetDeviceCount(&n_visible));
    for (int k = 0; k < n_devices; k++)
        if (devices[k] < 0 || devices[k] >= n_visible)
            return fail("rt_render_multi: devices[" + std::to_string(k) + "] = " + std::to_string(devices[k]) + " but " +
                        std::to_string(n_visible) + " device(s) are visible");
    const bool fixed = (flags & RT_FLAG_DETERMINISTIC) != 0;
    const uint32_t shard_flags = (flags & ~kFlagFixedFb) | (fixed ? kFlagFixedFb : 0u);
    const size_t n_values = 3 * (size_t)width * height;
    const size_t sum_bytes = n_values * (fixed ? sizeof(long long) : sizeof(float));
    int caller_device = 0;
    HIP_TRY(hipGetDevice(&caller_device));
    // every device's copy of the scene, before any thread starts (replicas are created under the scene's lock)
    std::vector<const rt_scene *> on_dev(n_devices, nullptr);
    for (int k = 0; k < n_devices; k++) {
        on_dev[k] = scene_on_device(scene, devices[k]);
        if (!on_dev[k]) return 1;
    }
    // device buffers: cached per (device, slot) like rt_render's; slot 2 + 2k = shard k's sums on its device, 3 + 2k = its
    // staging copy on devices[0], slot 1 = the post-processed image (fixed-point mode)
    std::vector<int> distinct(devices, devices + n_devices);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    if (distinct.back() >= kMaxDevices) return fail("rt_render_multi: device ordinal out of range");
    std::vector<std::unique_lock<std::mutex>> out_locks;  // (ascending device order: two concurrent calls cannot deadlock)
    for (int d : distinct) out_locks.emplace_back(g_dev_busy[d]);
    struct Home {  // the calling thread's device is restored on every return path
        int device;
        ~Home() { (void)hipSetDevice(device); }
    } home{caller_device};
    struct {
        void *alloc(int device, int slot, size_t bytes) { return hipSetDevice(device) == hipSuccess ? out_buffer(device, slot, bytes) : nullptr; }
    } buf;
    const int dev0 = devices[0];
    // Peer access between devices[0] and every other listed device, once per pair and process: with it the shards' sums travel
    // over xGMI straight into devices[0]'s memory; without it hipMemcpyPeerAsync still works, staged through host memory by
    // the runtime.  (Nothing of this has run on two PHYSICAL devices yet -- one-GPU boxes list a device twice; the first
    // multi-GPU run is the driver's scaling run, where bench.py's probe records what happened here: `peer_access`.)
    {
        static std::mutex peer_mutex;
        static std::vector<std::pair<int, int>> peer_done;
        std::lock_guard<std::mutex> peer_lock(peer_mutex);
        for (int k = 1; k < n_devices; k++) {
            const int dk = devices[k];
            if (dk == dev0 || std::find(peer_done.begin(), peer_done.end(), std::make_pair(dev0, dk)) != peer_done.end()) continue;
            peer_done.push_back({dev0, dk});
            int can01 = 0, can10 = 0;
            std::string why;
            if (hipDeviceCanAccessPeer(&can01, dev0, dk) != hipSuccess || hipDeviceCanAccessPeer(&can10, dk, dev0) != hipSuccess) why = "hipDeviceCanAccessPeer failed";
            else if (!can01 || !can10) why = "the devices report no peer access";
            else {
                hipError_t e1 = hipSetDevice(dev0) == hipSuccess ? hipDeviceEnablePeerAccess(dk, 0) : hipErrorInvalidDevice;
                hipError_t e2 = hipSetDevice(dk) == hipSuccess ? hipDeviceEnablePeerAccess(dev0, 0) : hipErrorInvalidDevice;
                if (e1 == hipErrorPeerAccessAlreadyEnabled) e1 = hipSuccess;  // (PyTorch, or an earlier library in the process, got there first)
                if (e2 == hipErrorPeerAccessAlreadyEnabled) e2 = hipSuccess;
                (void)hipGetLastError();
                if (e1 != hipSuccess || e2 != hipSuccess) why = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2);
            }
            g_peer_log += "devices " + std::to_string(dev0) + " <-> " + std::to_string(dk) + ": " + (why.empty() ? "peer access enabled" : "NO peer access (" + why + "): copies go through host memory") + "; ";
            if (!why.empty())
                fprintf(stderr, "rtcuda_amd: rt_render_multi: no peer access between devices %d and %d (%s): the shard's sums are copied through host memory\n", dev0, dk, why.c_str());
        }
        (void)hipSetDevice(caller_device);
    }
    // shard k renders into its own raw-sum buffer on ITS device; shards 1.. land in a staging buffer on devices[0]
    std::vector<void *> d_sum(n_devices, nullptr), d_stage(n_devices, nullptr);
    for (int k = 0; k < n_devices; k++) {
        d_sum[k] = buf.alloc(devices[k], 2 + 2 * k, sum_bytes);
        if (!d_sum[k]) return fail("rt_render_multi: out of device memory on device " + std::to_string(devices[k]));
        if (k > 0) {
            d_stage[k] = devices[k] == dev0 ? d_sum[k] : buf.alloc(dev0, 3 + 2 * k, sum_bytes);  // (same device: the buffer is its own staging)
            if (!d_stage[k]) return fail("rt_render_multi: out of device memory on device " + std::to_string(dev0));
        }
    }
    float *d_out = fixed ? (float *)buf.alloc(dev0, 1, n_values * sizeof(float)) : (float *)d_sum[0];
    if (!d_out) return fail("rt_render_multi: out of device memory");
    // ---- one host thread per device (render.cuh's render() is one thread on one device: this is the multi-device form of
    // the same call): select the device, zero the shard's sums, render slot shard k of n, hand the sums to devices[0]
    std::vector<rt_stats> sub(n_devices);
    std::vector<int> rc(n_devices, 0);
    std::vector<std::string> err(n_devices);
    std::vector<std::thread> th;
    for (int k = 0; k < n_devices; k++)
        th.emplace_back([&, k] {
            auto bail = [&](const char *what) { rc[k] = 1; err[k] = std::string("rt_render_multi: ") + what + " (device " + std::to_string(devices[k]) + ")"; };
            if (hipSetDevice(devices[k]) != hipSuccess) return bail("hipSetDevice failed");
            hipStream_t st;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return bail("stream create failed");
            if (hipMemsetAsync(d_sum[k], 0, sum_bytes, st) != hipSuccess) {
                bail("memset failed");
            } else {
                rc[k] = render_shard_impl(on_dev[k], camera, width, height, num_samples, max_bounces, seed, k, n_devices, shard_flags,
                                          (float *)d_sum[k], st, &sub[k], /* a context of its own per shard: */ 8 + k);
                if (rc[k]) err[k] = g_last_error;
                else if (k > 0 && d_stage[k] != d_sum[k]) {
                    const hipError_t pe = hipMemcpyPeerAsync(d_stage[k], dev0, d_sum[k], devices[k], sum_bytes, st);
                    if (pe != hipSuccess) bail((std::string("peer copy of the shard's sums failed: ") + hipGetErrorString(pe)).c_str());
                }
            }
            if (hipStreamSynchronize(st) != hipSuccess && rc[k] == 0) bail("stream synchronise failed");
            (void)hipStreamDestroy(st);
        });
    for (auto &t : th) t.join();
    for (int k = 0; k < n_devices; k++)
        if (rc[k]) return fail(err[k]);
    // ---- on devices[0]: add the shards' sums in shard order (a fixed order: the result does not depend on which device
    // finished first), post-process (render.cuh:330-338), copy out
    HIP_TRY(hipSetDevice(dev0));
    const int nv = (int)n_values;
    for (int k = 1; k < n_devices; k++) {
        if (fixed) hipLaunchKernelGGL(k_accumulate_i64, dim3((nv + 255) / 256), dim3(256), 0, nullptr, (long long *)d_sum[0], (const long long *)d_stage[k], nv);
        else hipLaunchKernelGGL(k_accumulate_f32, dim3((nv + 255) / 256), dim3(256), 0, nullptr, (float *)d_sum[0], (const float *)d_stage[k], nv);
    }
    HIP_TRY(hipGetLastError());
    int prc = fixed ? rt_post_process_fixed((const int64_t *)d_sum[0], d_out, width * height, num_samples, nullptr)
                    : rt_post_process(d_out, width * height, num_samples, nullptr);
    if (prc) return prc;
    HIP_TRY(hipMemcpy(out_rgb, d_out, n_values * sizeof(float), hipMemcpyDeviceToHost));
    if (stats) {
        rt_stats tot = sub[0];
        for (int k = 1; k < n_devices; k++) {
            tot.camera_rays += sub[k].camera_rays;
            tot.shade_events += sub[k].shade_events;
            tot.closest_rays += sub[k].closest_rays;
            tot.any_rays += sub[k].any_rays;
            tot.emission_adds += sub[k].emission_adds;
            tot.shadow_adds += sub[k].shadow_adds;
            tot.rr_draws += sub[k].rr_draws;
            tot.iterations = std::max(tot.iterations, sub[k].iterations);
            tot.launches_trace += sub[k].launches_trace;
            tot.seconds_trace = std::max(tot.seconds_trace, sub[k].seconds_trace);
            tot.seconds_advance = std::max(tot.seconds_advance, sub[k].seconds_advance);
            tot.seconds_render = std::max(tot.seconds_render, sub[k].seconds_render);  // the devices render side by side
            tot.seconds_rng_init = std::max(tot.seconds_rng_init, sub[k].seconds_rng_init);
            tot.seconds_reference_tree = std::max(tot.seconds_reference_tree, sub[k].seconds_reference_tree);
            for (int q = 4; q < 7; q++) tot.reserved[q] += sub[k].reserved[q];
        }
        tot.reserved[3] = n_devices;
        *stats = tot;
    }
    return 0;
}

int rt_trace_closest(const rt_scene *scene, int n, const float *origin_xyz, const float *dir_xyz, const float *tmax,
                     int32_t *hit_tri, float *t, float *u, float *v) {
    return rt_trace_closest_flags(scene, 0u, n, origin_xyz, dir_xyz, tmax, hit_tri, t, u, v);
}

int rt_trace_closest_flags(const rt_scene *scene, uint32_t flags, int n, const float *origin_xyz, const float *dir_xyz,
                           const float *tmax, int32_t *hit_tri, float *t, float *u, float *v) {
    if (!scene || n < 0 || (n > 0 && (!origin_xyz || !dir_xyz || !tmax || !hit_tri || !t || !u || !v)))
        return fail("rt_trace_closest: bad argument");
    if (n == 0) return 0;
    const bool literal = (flags & RT_FLAG_REFERENCE_WALK) != 0;
    const bool verify = !literal && (flags & RT_FLAG_WATERTIGHT) == 0;
    if ((literal || verify) && ensure_ref_tree(scene)) return 1;
    float *d_o, *d_d, *d_tm, *d_t, *d_u, *d_v;
    int *d_h;
    unsigned long long *d_vstat;
    DevScope tmp;
    if (tmp.alloc(d_vstat, 4)) return 1;
    HIP_TRY(hipMemset(d_vstat, 0, 4 * sizeof(unsigned long long)));
    if (tmp.alloc(d_o, 3 * (size_t)n) || tmp.alloc(d_d, 3 * (size_t)n) || tmp.alloc(d_tm, (size_t)n) || tmp.alloc(d_t, (size_t)n) ||
        tmp.alloc(d_u, (size_t)n) || tmp.alloc(d_v, (size_t)n) || tmp.alloc(d_h, (size_t)n))
        return 1;
    {
        float need[3] = {0.f, 0.f, 0.f};  // the 4-wide records must be padded for these origins (ensure_origin_radius)
        for (int i = 0; i < n; i++)
            for (int a = 0; a < 3; a++)
                if (std::isfinite(origin_xyz[3 * (size_t)i + a])) need[a] = std::max(need[a], std::fabs(origin_xyz[3 * (size_t)i + a]));
        if (int rc = ensure_origin_radius(scene, need)) return rc;
    }
    HIP_TRY(hipMemcpy(d_o, origin_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, dir_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tm, tmax, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
    const int test_grid = std::min(grid_for(n), 2048);
    const int stack_cap = lds_stack_cap(scene, kLdsStack);
    int *d_over = nullptr;
    int over_levels = 0;
    if (ensure_overflow(d_over, over_levels, std::max(scene->stack_bound, 32) - stack_cap)) return 1;
    tmp.ptrs.push_back(d_over);
    {
        TraceParams tp{};
        tp.total = n;
        tp.o3 = d_o;
        tp.d3 = d_d;
        tp.tmax = d_tm;
        tp.order = scene->d_order;
        tp.out_i = d_h;
        tp.out_t = d_t;
        tp.out_u = d_u;
        tp.out_v = d_v;
        tp.vstat = d_vstat;
        DPools none{};
        RT_LAUNCH_TRACE_REF(MODE_TEST_CLOSEST, literal, verify, scene->wide, dim3(test_grid), sizeof(int) * kBlock * (size_t)(stack_cap + 2), nullptr,
                        scene->dev(), none, tp, stack_cap, d_over);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hit_tri, d_h, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t, d_t, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(u, d_u, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(v, d_v, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

int rt_trace_any(const rt_scene *scene, int n, const float *origin_xyz, const float *dir_xyz, const float *tmax,
                 const int32_t *excluded_tri, int32_t *occluded) {
    return rt_trace_any_flags(scene, 0u, n, origin_xyz, dir_xyz, tmax, excluded_tri, occluded);
}

int rt_trace_any_flags(const rt_scene *scene, uint32_t flags, int n, const float *origin_xyz, const float *dir_xyz,
                       const float *tmax, const int32_t *excluded_tri, int32_t *occluded) {
    if (!scene || n < 0 || (n > 0 && (!origin_xyz || !dir_xyz || !tmax || !excluded_tri || !occluded)))
        return fail("rt_trace_any: bad argument");
    if (n == 0) return 0;
    const bool literal = (flags & RT_FLAG_REFERENCE_WALK) != 0;
    const bool verify = !literal && (flags & RT_FLAG_WATERTIGHT) == 0;
    if ((literal || verify) && ensure_ref_tree(scene)) return 1;
    std::vector<int> excl(n);
    for (int i = 0; i < n; i++) {
        int e = excluded_tri[i];
        excl[i] = (e >= 0 && e < scene->n_tris) ? scene->h_inverse[e] : -1;
    }
    float *d_o, *d_d, *d_tm;
    int *d_e, *d_occ;
    unsigned long long *d_vstat;
    DevScope tmp;
    if (tmp.alloc(d_vstat, 4)) return 1;
    HIP_TRY(hipMemset(d_vstat, 0, 4 * sizeof(unsigned long long)));
    if (tmp.alloc(d_o, 3 * (size_t)n) || tmp.alloc(d_d, 3 * (size_t)n) || tmp.alloc(d_tm, (size_t)n) || tmp.alloc(d_e, (size_t)n) ||
        tmp.alloc(d_occ, (size_t)n))
        return 1;
    {
        float need[3] = {0.f, 0.f, 0.f};  // the 4-wide records must be padded for these origins (ensure_origin_radius)
        for (int i = 0; i < n; i++)
            for (int a = 0; a < 3; a++)
                if (std::isfinite(origin_xyz[3 * (size_t)i + a])) need[a] = std::max(need[a], std::fabs(origin_xyz[3 * (size_t)i + a]));
        if (int rc = ensure_origin_radius(scene, need)) return rc;
    }
    HIP_TRY(hipMemcpy(d_o, origin_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, dir_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tm, tmax, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_e, excl.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    const int test_grid = std::min(grid_for(n), 2048);
    const int stack_cap = lds_stack_cap(scene, kLdsStack);
    int *d_over = nullptr;
    int over_levels = 0;
    if (ensure_overflow(d_over, over_levels, std::max(scene->stack_bound, 32) - stack_cap)) return 1;
    tmp.ptrs.push_back(d_over);
    {
        TraceParams tp{};
        tp.total = n;
        tp.o3 = d_o;
        tp.d3 = d_d;
        tp.tmax = d_tm;
        tp.excluded = d_e;
        tp.out_i = d_occ;
        tp.vstat = d_vstat;
        DPools none{};
        RT_LAUNCH_TRACE_REF(MODE_TEST_ANY, literal, verify, scene->wide, dim3(test_grid), sizeof(int) * kBlock * (size_t)(stack_cap + 2), nullptr,
                        scene->dev(), none, tp, stack_cap, d_over);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(occluded, d_occ, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

int rt_xorwow_states(uint64_t seed, uint32_t first, uint32_t count, int draws, uint32_t *state6, float *uniforms) {
    if (count == 0) return 0;
    if (!state6 || draws < 0 || (draws > 0 && !uniforms)) return fail("rt_xorwow_states: bad argument");
    if ((uint64_t)first + count > (uint64_t)kW) return fail("rt_xorwow_states: subsequence range exceeds W");
    DPools p{};
    uint32_t *buf = nullptr, *d_state = nullptr, *d_jump = nullptr;
    float *d_uni = nullptr;
    DevScope tmp;
    if (tmp.alloc(buf, 6 * (size_t)count) || tmp.alloc(d_state, 6 * (size_t)count) || tmp.alloc(d_uni, (size_t)count * draws) ||
        tmp.alloc(d_jump, (size_t)20 * 800))
        return 1;
    p.n = (int)count;  // only the six RNG arrays are touched: place array A_RD at buf
    p.base = (float *)buf - (size_t)A_RD * count;
    HIP_TRY(hipMemcpy(d_jump, jump_powers().data(), sizeof(uint32_t) * 20 * 800, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rng_init, dim3(grid_for((int)count)), dim3(kBlock), 0, nullptr, p, (int)count, (int)first,
                       xorwow_seed(seed), d_jump);
    hipLaunchKernelGGL(k_test_draw, dim3(grid_for((int)count)), dim3(kBlock), 0, nullptr, p, (int)count, draws, d_state,
                       d_uni);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(state6, d_state, sizeof(uint32_t) * 6 * (size_t)count, hipMemcpyDeviceToHost));
    if (draws > 0) HIP_TRY(hipMemcpy(uniforms, d_uni, sizeof(float) * (size_t)count * draws, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
