/* rtcuda_amd.h -- C-ABI of the MI355X-native render path (drop-in for lashhw/rtcuda's render()).
 *
 * Plain C: opaque handles, plain pointers and sizes, int status codes.  No torch / HIP types.
 * Every entry point names the reference interface it replaces (file:line into the reference
 * tree).  The C++ host API that keeps the reference's class names (Vec3 / Triangle / Material /
 * Light / Primitive / Bvh / Scene / Camera / render) is include/rtcuda/rtcuda.hpp -- a
 * header-only layer over exactly these functions.
 *
 * Conventions
 *   - return 0 on success, non-zero on failure; rt_last_error() gives the message
 *     (the reference prints and exit()s instead: utility.cuh:6-13).
 *   - all device memory is owned by the library behind rt_scene* / an internal per-device
 *     context (the reference leaks every allocation: main.cu:50,121,136; bvh.cuh:211-217;
 *     render.cuh:374-391).
 *   - "current device" is the calling thread's current HIP device (hipSetDevice /
 *     torch.cuda.set_device); one process per GPU is the intended deployment.
 *   - triangle, material and light POINTERS of the reference API become INDICES here.
 */
#ifndef RTCUDA_AMD_H
#define RTCUDA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_NUM_WORKING_PATHS 1048576 /* constant.hpp:8 -- number of path slots == RNG streams */

enum { RT_MATTE = 0, RT_MIRROR = 1, RT_GLASS = 2 };  /* material.cuh:4-8 */
enum { RT_POINT_LIGHT = 0, RT_AREA_LIGHT = 1 };      /* light.cuh:4-7 */

/* material.cuh:20-22 (same field order, 20 bytes) */
typedef struct rt_material {
    float albedo[3];
    float index_of_refraction;
    int32_t type;
} rt_material;

/* light.cuh:20-26 with `Triangle *d_triangle` flattened to a triangle index (32 bytes) */
typedef struct rt_light {
    int32_t type;
    float pos[3];     /* point light */
    int32_t triangle; /* area light: index into the scene's triangle array */
    float L[3];       /* radiance (area) or intensity I (point) */
} rt_light;

/* camera.cuh:11-14 (48 bytes): what Camera's constructor leaves in the object */
typedef struct rt_camera {
    float lookfrom[3];
    float upper_left[3];
    float horizontal[3];
    float vertical[3];
} rt_camera;

typedef struct rt_scene rt_scene;

/* Per-render counters (the reference has only stdout).  All counts are for THIS shard. */
typedef struct rt_stats {
    int64_t camera_rays;     /* gen events      (render.cuh:250)  */
    int64_t shade_events;    /* mat events      (render.cuh:139)  */
    int64_t closest_rays;    /* rays traced by the closest-hit kernel */
    int64_t any_rays;        /* rays traced by the any-hit kernel */
    int64_t emission_adds;   /* bounce-0 emission deposits (render.cuh:98-103) */
    int64_t shadow_adds;     /* unoccluded NEE deposits    (render.cuh:291-293) */
    int64_t rr_draws;        /* Russian-roulette draws     (render.cuh:117) */
    int64_t iterations;      /* stage rounds launched */
    int64_t bvh_nodes;       /* node records of the device BVH */
    int64_t bvh_depth;
    double seconds_render;   /* device time of the render loop (HIP events), excl. RNG init */
    double seconds_rng_init; /* device time of the one-off XORWOW state initialisation */
    double seconds_trace;    /* trace kernel (closest-hit + any-hit rays of a round in one launch): average
                                launch duration (HIP events on the launch stream, every 4th round sampled)
                                x launches; 0 unless RT_FLAG_TIME_KERNELS */
    double seconds_reference_tree; /* host seconds this call spent building + uploading the reference's own tree (first render of
                                      a scene without RT_FLAG_WATERTIGHT; 0 afterwards): a one-off like the BVH build */
    double seconds_advance;  /* same for the advance kernel */
    int64_t launches_trace;  /* launches of each stage kernel (= iterations) */
    int64_t reserved[7];     /* reserved[0] = launches actually sampled by the event timer;
                                reserved[1] = 1 when the frame ran as one persistent k_paths launch (then
                                seconds_trace is that launch's duration and launches_trace is 1);
                                reserved[2] = BVH node records that launch staged in LDS (small shards only);
                                reserved[3] = rt_render_multi: number of device shards behind these totals;
                                default kernels (no RT_FLAG_WATERTIGHT): reserved[4] = rays re-traced through the reference's
                                own tree, reserved[5] = accepted hits the reference's box test loses, reserved[6] = closest
                                hits with an exact tie at the final distance */
} rt_stats;

/* Flags for rt_render / rt_render_shard */
#define RT_FLAG_TIME_KERNELS 1u  /* time the stage kernels with HIP events on the launch stream (fills seconds_*) */
#define RT_FLAG_DETERMINISTIC 2u /* rt_render: accumulate in 64-bit fixed point (2^-30) instead of float atomics
                                    (vec3.cuh:149-153): bit-reproducible image, independent of summation order */

#define RT_FLAG_RNG_PER_SAMPLE 4u /* NOT the reference's random numbers: every camera ray starts a stream of its own, keyed by
                                    (seed, camera ray id), instead of continuing its path SLOT's stream (render.cuh:72,156,263).
                                    The image is a statistically equivalent estimate, not the reference's image sample for
                                    sample -- parity tests never use it.  What it buys: the frame no longer depends on which
                                    slot or GPU serves a camera ray, so with rt_render_shard every rank runs ALL W slots on
                                    num_samples / shard_count samples of every pixel (num_samples % shard_count == 0), the
                                    shards' fixed-point sums add up to the 1-GPU sums exactly, and 8 GPUs are not held to
                                    1/8 of the W chains each (SURVEY.md section 7 "per_sample", section 8b `rng_mode`) */

/* WHICH HITS A RAY FINDS: the reference's own decisions are the default.
 * Two results of lashhw/rtcuda are properties of its own BVH, not of the scene: its fp32 slab test on exact boxes
 * (aabb_intersector.cuh:14-36, hit iff entry <= exit, no look at tmax) drops about one accepted hit in 10^7 rays, and among
 * hits at exactly equal t the triangle its walk tests last wins (triangle.cuh:49).  Both are functions of the ray alone --
 * a triangle is visible to the reference's walk iff the box of its LEAF passes that slab test (the boxes above it are nested
 * exactly and fp32 rounding is monotone, so they pass with it) -- which lets the library reproduce them on its OWN tree:
 *
 *   flags = 0 (default; what bench.py times): the product's 4-wide walk; a shadow ray's occluder counts only if the
 *       reference's walk can see it; a path ray's closest hit is checked once (visible? no exact tie at the final distance?)
 *       and the ~2 rays in 10^7 that fail are re-traced through the reference's own binary SAH tree (bvh.cuh:30-219, built
 *       from the scene's triangles on first use).  The image equals the reference ALGORITHM's image ray for ray: every
 *       event total and every fixed-point pixel sum of the six full BASELINE frames equals the literal CPU restatement's.
 *   RT_FLAG_REFERENCE_WALK: every ray walks the reference's own tree as Bvh::traverse does (bvh.cuh:221-357).  Same image as
 *       the default, five times slower: the cross-check of the default, kept for that.
 *   RT_FLAG_WATERTIGHT: the triangle-list definition instead -- an accepted hit is never lost to a box test, ties go to the
 *       larger caller index: what exhaustive search over all triangles returns.  About 1 path in 4 * 10^6 differs from the
 *       reference's image (on BASELINE config 5 those paths carry whole light deposits: RMS 3.2e-4); 3 - 4 % faster.
 *
 * CAVEAT ("the reference" = its algorithm in separately rounded fp32): this library and the CPU oracle are built with
 * -ffp-contract=off, so inv * bound + scaled_origin is a multiplication and an addition.  The reference's CMake build uses
 * nvcc's defaults (fmad on): a CUDA binary contracts that expression -- and others -- into FMAs and loses a DIFFERENT handful
 * of rays; its libdevice sincosf / powf differ from the pinned forms here as well.  Bit parity with a CUDA binary is unpinned
 * and cannot be had offline; what is exact is parity with the literal restatement of the source. */
#define RT_FLAG_REFERENCE_WALK 8u /* every ray through the reference's own tree (see above); not with RT_FLAG_RNG_PER_SAMPLE */
#define RT_FLAG_WATERTIGHT 16u    /* the triangle-list definition of the hits (see above); not with RT_FLAG_REFERENCE_WALK */

/* ---- scene -------------------------------------------------------------------------------
 * Replaces: Triangle(p0,p1,p2) x n (triangle.cuh:6-7), cudaMalloc/Memcpy of triangles,
 * materials and lights (main.cu:50-51,119-122,136-137), Primitive(tri*,mat*,light*)
 * (primitive.cuh:6-7), Bvh(triangles, primitives) (bvh.cuh:30-219) and the Scene aggregate
 * (scene.cuh:4-8).  tri_p0p1p2 is n_tris x 9 floats; tri_light[i] is the index into `lights`
 * of the area light that triangle i carries, or -1 (may be NULL = no area lights).
 * The light ORDER is the caller's (the reference's comes from unordered_map iteration,
 * main.cu:128).  Uploads to the current device. */
int rt_scene_create(const float *tri_p0p1p2, int n_tris, const int32_t *tri_material,
                    const int32_t *tri_light, const rt_material *materials, int n_materials,
                    const rt_light *lights, int n_lights, rt_scene **out_scene);
void rt_scene_destroy(rt_scene *scene);

/* out[0]=node records, out[1]=triangles, out[2]=max depth, out[3]=leaves */
int rt_scene_info(const rt_scene *scene, int64_t out[4]);
/* Which BVH builder made the scene (0 = host SAH, the default; 1 = device LBVH, RT_BVH_BUILDER=lbvh) and
 * how long the build took (host wall clock / HIP events).  The reference times "Top-down constructing BVH"
 * on stdout (bvh.cuh:106-201). */
int rt_scene_build_info(const rt_scene *scene, int *builder, double *seconds);

/* Replaces Camera::Camera(lookfrom, lookat, up, vfov_deg, aspect) (camera.cuh:15-29). Host only. */
int rt_camera_make(const float lookfrom[3], const float lookat[3], const float up[3], float vfov_deg,
                   float aspect_ratio, rt_camera *out);

/* ---- render ------------------------------------------------------------------------------
 * Replaces render(width, height, num_samples, max_bounces, camera, scene, framebuffer)
 * (render.cuh:366-457) with RAND_SEED (render.cuh:417) exposed as `seed` (reference: 1).
 * out_rgb: HOST buffer of width*height*3 floats, row-major, top row first, each channel
 * sqrt(sum/spp) exactly as post_process_framebuffer leaves it (render.cuh:330-338). */
int rt_render(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
              int max_bounces, uint64_t seed, uint32_t flags, float *out_rgb, rt_stats *stats);

/* The same call over several GPUs of one node, in ONE process (no reference counterpart: render.cuh:366-367 drives one
 * device from one thread).  devices[0 .. n_devices) are HIP device ordinals; n_devices must divide W (1, 2, 4, 8, ...).
 * The scene is replicated onto every listed device on first use (the replicas belong to `scene` and die with it), one
 * host thread per device renders slot shard k of n_devices (see rt_render_shard) into a raw-sum buffer on its own
 * device, the shards' sums are copied to devices[0] (peer copies over xGMI), added there in shard order, post-processed
 * and copied to out_rgb (HOST, as rt_render).  A device may be listed more than once (its shards then run side by side
 * on it) -- which is also how the path is tested on a one-GPU box: THE COPY BETWEEN TWO PHYSICAL DEVICES HAS NOT RUN YET (see
 * rt_peer_access_log).  Peer access devices[0] <-> devices[k] is enabled on first use where the devices offer it; without it
 * the copies go through host memory.  With RT_FLAG_DETERMINISTIC the result is bit-equal
 * to rt_render's with the same flag, whatever the device list.  stats: counts summed over the shards, times the
 * maximum over the shards, reserved[3] = n_devices.  The calling thread's current device is left as it was.
 * (The process-per-GPU deployment -- rt_render_shard under torch.distributed / RCCL, bench.py -- is the other way to
 * the same image; this entry point is for a C++ driver that wants all GPUs behind one call.) */
int rt_render_multi(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
                    int max_bounces, uint64_t seed, uint32_t flags, const int *devices, int n_devices, float *out_rgb,
                    rt_stats *stats);

/* Multi-GPU building block: render only the camera rays owned by path slots
 * [shard_index*W/shard_count, (shard_index+1)*W/shard_count) -- slot s serves exactly the camera
 * rays c with c % W == s, so shards are disjoint and their raw sums add up to the 1-GPU image.
 * d_sum_rgb: DEVICE buffer of width*height*3 floats on the current device; contributions are
 * ADDED to it (zero it first).  stream: hipStream_t (NULL = default stream).  The call is
 * synchronous on that stream when it returns.  shard_count must divide W.
 * (Every render entry point: the scene's BVH records are padded for ray origins within the scene's own bounds; the first
 * render from a camera whose lookfrom lies outside them widens the padding and uploads the records again, once, a few
 * milliseconds.  Thread-safe; the image does not depend on it.) */
int rt_render_shard(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
                    int max_bounces, uint64_t seed, int shard_index, int shard_count, uint32_t flags,
                    float *d_sum_rgb, void *stream, rt_stats *stats);

/* Order-independent variant of rt_render_shard: d_sum_fixed is a DEVICE buffer of width*height*3 int64
 * fixed-point sums (units of 2^-30), ADDED to.  Integer adds commute, so the sums of the shards of a
 * multi-GPU render add up to EXACTLY the single-GPU sums and every run gives the same bits (the
 * reference's float atomicAdd, vec3.cuh:149-153, does not).  rt_post_process_fixed converts to the
 * post-processed float image: c = sqrt(float(sum * 2^-30) * (1/spp)). */
int rt_render_shard_fixed(const rt_scene *scene, const rt_camera *camera, int width, int height, int num_samples,
                          int max_bounces, uint64_t seed, int shard_index, int shard_count, uint32_t flags,
                          int64_t *d_sum_fixed, void *stream, rt_stats *stats);
int rt_post_process_fixed(const int64_t *d_sum_fixed, float *d_rgb_out, int num_pixels, int num_samples, void *stream);

/* post_process_framebuffer (render.cuh:330-338) on a DEVICE buffer: c = sqrt(c * (1/spp)). */
int rt_post_process(float *d_rgb, int num_pixels, int num_samples, void *stream);

/* ---- stage-level entry points (parity tests call these; HOST pointers, AoS xyz triples) -----
 * Closest hit (Bvh::traverse, bvh.cuh:251-303; ch(), render.cuh:297-328): hit_tri = index of
 * the hit triangle in the caller's ORIGINAL order or -1; t,u,v as Intersection
 * (intersection.hpp:4-6), undefined on a miss. */
int rt_trace_closest(const rt_scene *scene, int n, const float *origin_xyz, const float *dir_xyz,
                     const float *tmax, int32_t *hit_tri, float *t, float *u, float *v);
/* Any hit excluding one triangle (bvh.cuh:306-357; ah(), render.cuh:278-294): occluded[i] in {0,1}. */
int rt_trace_any(const rt_scene *scene, int n, const float *origin_xyz, const float *dir_xyz,
                 const float *tmax, const int32_t *excluded_tri, int32_t *occluded);
/* The same two entry points with a flags word (RT_FLAG_REFERENCE_WALK / RT_FLAG_WATERTIGHT: see the flags).  flags = 0 is
 * rt_trace_closest / rt_trace_any: hit_tri, t, u, v and occluded are the reference's answers, ties and lost hits included.
 * Directions are unit vectors (finite, every component below 2^126 in magnitude). */
int rt_trace_closest_flags(const rt_scene *scene, uint32_t flags, int n, const float *origin_xyz, const float *dir_xyz,
                           const float *tmax, int32_t *hit_tri, float *t, float *u, float *v);
int rt_trace_any_flags(const rt_scene *scene, uint32_t flags, int n, const float *origin_xyz, const float *dir_xyz,
                       const float *tmax, const int32_t *excluded_tri, int32_t *occluded);
/* curand_init(seed, subsequence, 0) for subsequences [first, first+count) (render.cuh:68-73):
 * state6 receives count x {d, v0..v4}.  Then `draws` uniforms per state into uniforms
 * (count x draws, may be 0 / NULL), advancing the returned states. */
int rt_xorwow_states(uint64_t seed, uint32_t first, uint32_t count, int draws, uint32_t *state6,
                     float *uniforms);

/* Releases every device allocation the library holds behind the scenes: the per-device render contexts (path pools, RNG
 * states, counters, overflow stacks, events) and the cached output buffers of rt_render / rt_render_multi.  Scenes are the
 * caller's (rt_scene_destroy).  No render may be in flight; the library keeps working afterwards (everything is re-created on
 * demand).  The reference frees nothing at all (render.cuh:374-391, bvh.cuh:211-217). */
void rt_shutdown(void);

/* rt_render_multi: what became of peer access between devices[0] and every other listed device, pair by pair, since the process
 * started ("devices 0 <-> 1: peer access enabled; ..." or the reason it is not and that copies go through host memory).  Empty
 * until a call has listed two different devices.  NOTE: that path has not run on two PHYSICAL GPUs yet (one-GPU boxes list a
 * device twice); bench.py's multi-GPU probe records this string so that a first real run can be diagnosed from its output. */
const char *rt_peer_access_log(void);

const char *rt_last_error(void);
const char *rt_version(void);
/* Hash of the sources, compiler flags and experiment defines this library's device code was built from (csrc/Makefile).
 * Measurement artefacts (profiles/pmc_k_paths.json) carry it, so counters are never priced against another build. */
const char *rt_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* RTCUDA_AMD_H */
