// rt_host_check.cpp -- host-only checks of the BVH builder (rt_bvh.h) and a CPU walk of the product's tree, g++.
//
// Runs without a GPU:
//   * rt_bvh_selfcheck  validates both record formats the kernels can be given -- `quads` (4-wide, the default: two
//     pair-style records per node) and `pairs` (2-wide) -- structurally (every triangle in exactly one leaf, every node
//     reachable exactly once, every record box contains what lies beneath it, stack bound) and walks both on the CPU
//     with the control flow and the fp32 expressions of inner_step / the triangle blocks in rtcuda_amd.hip, comparing
//     with an exhaustive search.  A malformed tree would hang or fault the GPU; this is where it is caught first.
//   * rt_hostwalk_*     the same walk for arbitrary rays (closest hit / any hit), with work counters: the traversal
//     audit (tests/test_traversal_audit.py) replays the rays of an oracle render through it.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "rt_bvh.h"
#include "rt_ref_tree.h"

namespace {
struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct Tri { V3 p0, e1, e2, n; };
inline bool tri_hit(const Tri &tr, V3 o, V3 d, float tmax, float &t) {
    V3 c = sub(tr.p0, o);
    V3 r = cross(d, c);
    float inv_det = 1.f / dot(d, tr.n);
    float u = inv_det * dot(tr.e2, r);
    float v = inv_det * dot(tr.e1, r);
    if (u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f) {
        float tt = inv_det * dot(c, tr.n);
        if (0 < tt && tt <= tmax) { t = tt; return true; }
    }
    return false;
}
// (as box_hit / inner_step in rtcuda_amd.hip; the kernels' 1 / d is v_rcp_f32, 1 ulp, here an exact division)
inline bool box_hit(V3 o, V3 inv, const float *lo, const float *hi, float tmax, float &entry) {
    float ax = (lo[0] - o.x) * inv.x, bx = (hi[0] - o.x) * inv.x;
    float ay = (lo[1] - o.y) * inv.y, by = (hi[1] - o.y) * inv.y;
    float az = (lo[2] - o.z) * inv.z, bz = (hi[2] - o.z) * inv.z;
    float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    entry = t_in;
    t_out = t_out * 1.000001f;
    return t_in <= t_out && t_out >= 0.f && t_in <= tmax * 1.000001f;
}
// The 4-wide node step of the kernels: plane distance as ONE fma, b * (1 / d) + s with s = -o * (1 / d) rounded on its own
// (inner_step<true> in rtcuda_amd.hip); the records it is given are padded for that (rt_bvh.h, pad_quads_for_origins)
inline bool box_hit_fma(V3 o, V3 inv, const float *lo, const float *hi, float tmax, float &entry) {
    const float sx = -o.x * inv.x, sy = -o.y * inv.y, sz = -o.z * inv.z;
    float ax = fmaf(lo[0], inv.x, sx), bx = fmaf(hi[0], inv.x, sx);
    float ay = fmaf(lo[1], inv.y, sy), by = fmaf(hi[1], inv.y, sy);
    float az = fmaf(lo[2], inv.z, sz), bz = fmaf(hi[2], inv.z, sz);
    float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    entry = t_in;
    t_out = t_out * 1.000001f;
    return fmaxf(t_in, 0.f) <= fminf(t_out, fminf(tmax * 1.000001f, FLT_MAX));
}
inline V3 inv_dir(V3 d) {
    auto clampinv = [](float x) { return 1.f / ((fabsf(x) < FLT_EPSILON) ? copysignf(FLT_EPSILON, x) : x); };
    return V3{clampinv(d.x), clampinv(d.y), clampinv(d.z)};
}
std::vector<Tri> leaf_order_triangles(const float *verts, const rtbvh::Result &r, int n) {
    std::vector<Tri> tris(n);
    for (int k = 0; k < n; k++) {
        const float *q = verts + 9 * (size_t)r.order[k];
        V3 p0{q[0], q[1], q[2]}, p1{q[3], q[4], q[5]}, p2{q[6], q[7], q[8]};
        tris[k].p0 = p0; tris[k].e1 = sub(p0, p1); tris[k].e2 = sub(p2, p0); tris[k].n = cross(tris[k].e1, tris[k].e2);
    }
    return tris;
}

struct WalkResult {
    int best = -1;          // closest: leaf-order triangle index or -1
    float t = 0.f;
    bool occluded = false;  // any hit
    long long nodes = 0, tris = 0, leaves = 0;
    int max_sp = 0;
    long long sp_hist[32] = {0};  // node steps by the stack size the lane has when it makes them (>= 31 in the last bin)
    bool failed = false;
    bool unseen_occluder = false;  // verified any-hit walks: the occluder found is invisible to the reference's walk
    float tie_t = -1.f;     // closest: the distance of the last EXACT tie between two accepted hits (a tie at the final t
                            // is what the reference's tree order decides: triangle.cuh:49)
    long long own_fail = 0, leaf_fail = 0;  // verified walks: hits whose own box / whose reference leaf box fails the reference's slab test
};

// ---- the reference's slab test (aabb_intersector.cuh:14-36), as reference_walk / ref_visible in rtcuda_amd.hip
struct RefSlab {
    bool nx, ny, nz;
    V3 inv, so;
    bool neg_zero;  // a direction component is -0.0: the octant (d < 0: false) and the sign of 1 / d disagree
};
inline bool is_neg_zero(float x) { uint32_t b; memcpy(&b, &x, 4); return b == 0x80000000u; }
inline RefSlab ref_slab(V3 o, V3 d) {
    RefSlab s;
    s.nx = d.x < 0; s.ny = d.y < 0; s.nz = d.z < 0;
    s.inv = inv_dir(d);
    s.so = V3{(-o.x) * s.inv.x, (-o.y) * s.inv.y, (-o.z) * s.inv.z};
    s.neg_zero = is_neg_zero(d.x) || is_neg_zero(d.y) || is_neg_zero(d.z);
    return s;
}
// b = [xmin, xmax, ymin, ymax, zmin, zmax] (bounding_box.cuh:15)
inline bool ref_box(const RefSlab &s, const float *b, float &entry) {
    const float ex = s.inv.x * b[s.nx ? 1 : 0] + s.so.x;
    const float ey = s.inv.y * b[s.ny ? 3 : 2] + s.so.y;
    const float ez = s.inv.z * b[s.nz ? 5 : 4] + s.so.z;
    entry = fmaxf(ex, fmaxf(ey, ez));
    const float xx = s.inv.x * b[s.nx ? 0 : 1] + s.so.x;
    const float xy = s.inv.y * b[s.ny ? 2 : 3] + s.so.y;
    const float xz = s.inv.z * b[s.nz ? 4 : 5] + s.so.z;
    return entry <= fminf(xx, fminf(xy, xz));
}
// What the reference's walk can SEE.  Its box test does not look at tmax (aabb_intersector.cuh:35), so whether it ever
// reaches a triangle depends on the ray alone: every box on the way down to the triangle's leaf must pass.  Those
// boxes are nested exactly (a node's box is the min / max of its triangles' boxes, bvh.cuh:57-61,150-160), rounding is
// monotone, and a slab term inv * bound + so is a monotone function of the bound -- so (octants consistent with the
// signs of 1 / d, i.e. no -0.0 component) a parent's entry distance is <= its child's and its exit distance >= its
// child's: IF THE LEAF'S BOX PASSES, EVERY ANCESTOR'S PASSES.  The triangle's own box (triangle.cuh:22-37) lies inside
// the leaf's, so a pass on it is a pass on the leaf: the common case needs no memory access at all.
// The one case in which octant and sign of 1 / d disagree is a direction component of exactly -0.0: the nesting argument
// does not hold then, and the ancestors are tested one by one through the parent links.
struct RefView {
    rtref::Tree tree;
    std::vector<int> leaf_of;  // caller's triangle index -> node index of its leaf in the reference's tree
    std::vector<int> parent;   // node -> parent node (root: -1)
    bool root_leaf = true;
};
inline void own_box(const Tri &tr, float *b) {  // triangle.cuh:9-10,22-37 on the stored record
    const V3 p1 = sub(tr.p0, tr.e1), p2{tr.p0.x + tr.e2.x, tr.p0.y + tr.e2.y, tr.p0.z + tr.e2.z};
    b[0] = fminf(tr.p0.x, fminf(p1.x, p2.x)); b[1] = fmaxf(tr.p0.x, fmaxf(p1.x, p2.x));
    b[2] = fminf(tr.p0.y, fminf(p1.y, p2.y)); b[3] = fmaxf(tr.p0.y, fmaxf(p1.y, p2.y));
    b[4] = fminf(tr.p0.z, fminf(p1.z, p2.z)); b[5] = fmaxf(tr.p0.z, fmaxf(p1.z, p2.z));
}
inline bool ref_visible(const RefView &rv, const RefSlab &s, const Tri &tr, int caller_index, WalkResult &w) {
    float b[6], e;
    own_box(tr, b);
    if (ref_box(s, b, e) && !s.neg_zero) return true;
    w.own_fail++;
    if (rv.root_leaf) return true;  // bvh.cuh:252 / :307: a root that is a leaf is intersected without a box test
    int node = rv.leaf_of[(size_t)caller_index];
    bool vis = ref_box(s, rv.tree.nodes[(size_t)node].box.b, e);
    if (vis && s.neg_zero)
        for (node = rv.parent[(size_t)node]; vis && node > 0; node = rv.parent[(size_t)node])  // (the root's own box is never tested)
            vis = ref_box(s, rv.tree.nodes[(size_t)node].box.b, e);
    if (!vis) w.leaf_fail++;
    return vis;
}
// One ray through one of the two record formats.  wide: node = records cur, cur + 1 (children 0, 1 | 2, 3): the nearest
// child the ray may enter becomes the cursor, the others are pushed in record order; 2-wide: near child, far child
// pushed.  mode 0: closest hit (ties: the larger caller index, closest_hit_wins); mode 1: any hit but `excl`.
// `rv` (verified walks, mode 1): the first accepted hit ends the walk, as always; if the reference's walk cannot see its
// triangle, the result is flagged (`unseen_occluder`) and the caller asks the literal walk.
WalkResult walk_ray(const std::vector<rtbvh::Pair> &rec, bool wide, const std::vector<Tri> &tris, const std::vector<int32_t> &order,
                    int stack_entries, int mode, V3 o, V3 d, float tmax, int excl, const RefView *rv = nullptr) {
    WalkResult w;
    RefSlab slab{};
    if (rv) slab = ref_slab(o, d);
    std::vector<int> stack(stack_entries + 8);
    const V3 inv = inv_dir(d);
    int sp = 0, cur = tris.empty() ? rtbvh::kNoChild : 0;
    long long steps = 0;
    float t;
    while (cur != rtbvh::kNoChild && !w.occluded) {
        if (++steps > 1000000) { w.failed = true; break; }
        if (cur >= 0) {
            w.nodes++;
            w.sp_hist[sp < 31 ? sp : 31]++;
            const float *boxes[4];
            int links[4];
            int nk = 0;
            for (int half = 0; half < (wide ? 2 : 1); half++) {
                const rtbvh::Pair &p = rec[cur + half];
                boxes[nk] = p.lbox; links[nk++] = p.llink;
                boxes[nk] = p.rbox; links[nk++] = p.rlink;
            }
            float e[4];
            bool h[4];
            for (int k = 0; k < nk; k++)
                h[k] = (wide ? box_hit_fma(o, inv, boxes[k], boxes[k] + 3, tmax, e[k]) : box_hit(o, inv, boxes[k], boxes[k] + 3, tmax, e[k])) &&
                       links[k] != rtbvh::kNoChild;
            int near_k = -1;
            for (int k = 0; k < nk; k++)  // nearest entered child; ties: the lower index (as the kernels' !(a > b) selects)
                if (h[k] && (near_k < 0 || e[k] < e[near_k])) near_k = k;
            if (near_k < 0) {
                cur = sp > 0 ? stack[--sp] : rtbvh::kNoChild;
            } else {
                for (int k = 0; k < nk; k++)
                    if (h[k] && k != near_k) {
                        if (sp >= (int)stack.size()) { w.failed = true; break; }
                        stack[sp++] = links[k];
                    }
                if (w.failed) break;
                cur = links[near_k];
                if (sp > w.max_sp) w.max_sp = sp;
            }
        } else {
            int ref = ~cur, first = ref >> 3, count = ref & 7;
            w.leaves++;
            for (int k = first; k < first + count; k++) {
                w.tris++;
                if (tri_hit(tris[k], o, d, tmax, t)) {
                    if (mode == 1) {
                        if (k != excl) {
                            w.occluded = true;
                            w.unseen_occluder = rv && !ref_visible(*rv, slab, tris[k], order[k], w);
                            break;
                        }
                    } else {
                        if (t == tmax && w.best >= 0) w.tie_t = t;
                        if (!(t == tmax && w.best >= 0) || order[k] > order[w.best]) {  // closest_hit_wins()
                            tmax = t;
                            w.best = k;
                        }
                    }
                }
            }
            cur = sp > 0 ? stack[--sp] : rtbvh::kNoChild;
        }
    }
    w.t = w.best >= 0 ? tmax : 0.f;
    return w;
}

// The 4-wide records as the kernels are given them: padded for the origins of the rays at hand (the scene's own bounds, and
// whatever lies farther out in this batch -- ensure_origin_radius in rtcuda_amd.hip does the same per render / per call)
std::vector<rtbvh::Pair> padded_quads(const rtbvh::Result &r, int n_rays, const float *o3) {
    float radius[3];
    rtbvh::quads_abs_bounds(r.quads, radius);
    for (int i = 0; i < n_rays; i++)
        for (int a = 0; a < 3; a++) {
            const float v = o3[3 * (size_t)i + a];
            if (std::isfinite(v) && std::fabs(v) * 1.001f > radius[a]) radius[a] = 2.f * std::fabs(v) * 1.001f;
        }
    if (rtbvh::knob("RT_NO_ORIGIN_PAD")) radius[0] = radius[1] = radius[2] = 0.f;  // (tests: shows what the padding is for)
    std::vector<rtbvh::Pair> out;
    rtbvh::pad_quads_for_origins(r.quads, radius, out);
    return out;
}
// structural validation of one record format: returns the number of errors
int64_t validate(const rtbvh::Result &r, const std::vector<rtbvh::Pair> &rec, bool wide, const float *verts, int n, int &max_leaf) {
    const int per_node = wide ? 2 : 1;
    const int n_nodes = (int)rec.size() / per_node;
    int64_t errors = (int)rec.size() % per_node ? 1 : 0;
    std::vector<int> seen_tri(n, 0), seen_node(n_nodes, 0);
    if (n_nodes > 0) seen_node[0] = 1;
    for (int ni = 0; ni < n_nodes; ni++)
        for (int k = 0; k < 2 * per_node; k++) {
            const rtbvh::Pair &p = rec[per_node * ni + (k >> 1)];
            const int link = (k & 1) ? p.rlink : p.llink;
            const float *box = (k & 1) ? p.rbox : p.lbox;
            if (link == rtbvh::kNoChild) continue;
            if (link < 0) {
                int ref = ~link, first = ref >> 3, count = ref & 7;
                if (count <= 0 || first < 0 || first + count > n) { errors++; continue; }
                max_leaf = std::max(max_leaf, count);
                for (int q = first; q < first + count; q++) {
                    int ti = r.order[q];
                    if (ti < 0 || ti >= n) { errors++; continue; }
                    seen_tri[ti]++;
                    const float *v = verts + 9 * (size_t)ti;
                    for (int c = 0; c < 3; c++)
                        for (int a = 0; a < 3; a++)
                            if (!(v[3 * c + a] >= box[a]) || !(v[3 * c + a] <= box[3 + a])) errors++;
                }
            } else {
                if (link % per_node != 0 || link / per_node >= n_nodes || link / per_node <= ni) { errors++; continue; }  // children come later
                seen_node[link / per_node]++;
                for (int g = 0; g < 2 * per_node; g++) {  // every grandchild box lies inside this child box
                    const rtbvh::Pair &cp = rec[link + (g >> 1)];
                    if (((g & 1) ? cp.rlink : cp.llink) == rtbvh::kNoChild) continue;
                    const float *gb = (g & 1) ? cp.rbox : cp.lbox;
                    for (int a = 0; a < 3; a++)
                        if (gb[a] < box[a] - 1e-6f || gb[3 + a] > box[3 + a] + 1e-6f) errors++;  // (both padded by 2 ulps)
                }
            }
        }
    for (int i = 0; i < n; i++) if (seen_tri[i] != 1) errors++;
    for (int i = 0; i < n_nodes; i++) if (seen_node[i] != 1) errors++;
    return errors;
}
}  // namespace

extern "C" {
// out: [4-wide nodes, leaves, depth of the 4-wide tree, max_leaf_size, structural_errors, walk_mismatches, max_stack, max_steps
//       (node visits of the longest walk), stack_bound, binary_depth]
int rt_bvh_selfcheck(const float *verts, int n, int n_rays, const float *o3, const float *d3, int64_t *out10) {
    rtbvh::Result r = rtbvh::build(verts, n);
    memset(out10, 0, 10 * sizeof(int64_t));
    out10[0] = (int64_t)r.nodes.size();
    out10[1] = r.num_leaves;
    out10[2] = r.max_depth;
    out10[8] = r.stack_bound;
    out10[9] = r.bin_depth;
    int max_leaf = 0;
    int64_t errors = r.ok ? 0 : 1;
    if (r.quads.size() != 2 * r.nodes.size()) errors++;
    if (n > 0) {
        errors += validate(r, r.quads, true, verts, n, max_leaf);
        {   // the records as the kernels are given them (padded for the origins of this call's rays): the same structure, and
            // every plane at least 2^-24 x radius farther out than the builder's -- what the one-fma plane distance needs
            float radius[3];
            rtbvh::quads_abs_bounds(r.quads, radius);
            for (int i = 0; i < n_rays; i++)
                for (int a = 0; a < 3; a++)
                    if (std::isfinite(o3[3 * (size_t)i + a])) radius[a] = std::max(radius[a], std::fabs(o3[3 * (size_t)i + a]));
            const std::vector<rtbvh::Pair> padded = padded_quads(r, n_rays, o3);
            int ml = 0;
            errors += validate(r, padded, true, verts, n, ml);
            if (padded.size() != r.quads.size()) errors++;
            else if (!rtbvh::knob("RT_NO_ORIGIN_PAD"))
                for (size_t k = 0; k < padded.size(); k++)
                    for (int side = 0; side < 2; side++) {
                        if ((side ? padded[k].rlink : padded[k].llink) != (side ? r.quads[k].rlink : r.quads[k].llink)) errors++;
                        if ((side ? padded[k].rlink : padded[k].llink) == rtbvh::kNoChild) continue;
                        const float *pb = side ? padded[k].rbox : padded[k].lbox, *bb = side ? r.quads[k].rbox : r.quads[k].lbox;
                        for (int a = 0; a < 3; a++) {
                            const double need = std::ldexp((double)radius[a], -24);
                            if (!((double)bb[a] - (double)pb[a] >= need) || !((double)pb[3 + a] - (double)bb[3 + a] >= need)) errors++;
                        }
                    }
        }
        if (r.pairs.size() > 1 || r.pairs[0].llink != rtbvh::kNoChild) errors += validate(r, r.pairs, false, verts, n, max_leaf);
    }
    out10[3] = max_leaf;
    out10[4] = errors;
    if (errors) return 0;
    // CPU walks with the kernels' control flow vs exhaustive search, both formats
    const std::vector<Tri> tris = leaf_order_triangles(verts, r, n);
    int64_t mism = 0, max_stack = 0, max_steps = 0, mism_fmt[2] = {0, 0};
    double sum_nodes[2] = {0, 0}, sum_tris[2] = {0, 0};
    const std::vector<rtbvh::Pair> wide_recs = padded_quads(r, n_rays, o3);  // (as the kernels are given them)
    for (int i = 0; i < n_rays; i++) {
        V3 o{o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]}, d{d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]};
        float bt = FLT_MAX, t;
        int bb = -1;
        for (int k = 0; k < n; k++)
            if (tri_hit(tris[k], o, d, bt, t) && (!(t == bt && bb >= 0) || r.order[k] > r.order[bb])) { bt = t; bb = k; }
        for (int fmt = 0; fmt < 2; fmt++) {
            const bool wide = fmt == 0;
            WalkResult w = walk_ray(wide ? wide_recs : r.pairs, wide, tris, r.order, wide ? r.stack_bound : r.pair_depth + 1, 0, o, d, FLT_MAX, -1);
            if (w.failed) { mism += 1000000; continue; }
            if (w.best != bb || (bb >= 0 && w.t != bt)) { mism++; mism_fmt[fmt]++; }
            if (wide) { max_stack = std::max<int64_t>(max_stack, w.max_sp); max_steps = std::max<int64_t>(max_steps, w.nodes); }
            sum_nodes[fmt] += (double)w.nodes;
            sum_tris[fmt] += (double)w.tris;
        }
    }
    out10[5] = mism; out10[6] = max_stack; out10[7] = max_steps;
    if (getenv("RT_BVH_STATS") && n_rays > 0)
        fprintf(stderr, "bvh walk mismatches: 4-wide %lld, 2-wide %lld\n", (long long)mism_fmt[0], (long long)mism_fmt[1]);
    if (getenv("RT_BVH_STATS") && n_rays > 0)
        fprintf(stderr, "bvh walk per ray: 4-wide %.2f nodes %.2f tris | 2-wide %.2f nodes %.2f tris | %zu quads records, %zu pairs, depths %d / %d\n",
                sum_nodes[0] / n_rays, sum_tris[0] / n_rays, sum_nodes[1] / n_rays, sum_tris[1] / n_rays, r.quads.size(), r.pairs.size(), r.max_depth, r.pair_depth);
    return 0;
}

// ---- CPU walk of the product's tree for arbitrary rays.  The format follows the product's choice: 4-wide unless
// RT_BVH_WIDE=0 (rt_scene_create reads the same variable).
struct HostWalk {
    rtbvh::Result r;
    std::vector<Tri> tris;     // leaf order
    std::vector<int> inverse;  // original index -> leaf-order index
    bool wide = true;
    int n = 0;
    RefView ref;               // the reference's own tree (rt_ref_tree.h), for the verified walks
};
// Bvh::traverse (bvh.cuh:251-303 / :306-357) over the reference's tree, as reference_walk in rtcuda_amd.hip: left child
// before right child, a leaf child intersected on the spot, near inner child first by fp32 entry distance, the later
// tested triangle wins t <= tmax.  best / excl: leaf-order indices of the product (as everywhere in this file).
void literal_walk(const HostWalk &hw, int mode, V3 o, V3 d, float tmax, int excl, WalkResult &w) {
    const rtref::Tree &t = hw.ref.tree;
    w.best = -1;
    w.occluded = false;
    if (hw.n == 0) return;
    auto leaf = [&](const rtref::Node &nd) -> bool {
        for (int i = nd.link; i < nd.link + nd.count; i++) {
            const int k = hw.inverse[(size_t)t.prims[(size_t)i]];
            float tt;
            if (tri_hit(hw.tris[(size_t)k], o, d, tmax, tt)) {
                if (mode == 1) {
                    if (k != excl) { w.occluded = true; return true; }
                } else {
                    tmax = tt;
                    w.best = k;
                }
            }
        }
        return false;
    };
    if (t.nodes[0].count > 0) {
        leaf(t.nodes[0]);
    } else {
        const RefSlab s = ref_slab(o, d);
        int stk[64], sp = 0, left = t.nodes[0].link;
        while (true) {
            const rtref::Node &a = t.nodes[(size_t)left], &b = t.nodes[(size_t)left + 1];
            float el, er;
            bool gl = ref_box(s, a.box.b, el);
            if (gl && a.count > 0) { if (leaf(a)) break; gl = false; }
            bool gr = ref_box(s, b.box.b, er);
            if (gr && b.count > 0) { if (leaf(b)) break; gr = false; }
            if (gl && gr) {
                const bool right_first = el > er;
                stk[sp++] = right_first ? a.link : b.link;
                left = right_first ? b.link : a.link;
            } else if (gl) left = a.link;
            else if (gr) left = b.link;
            else { if (sp == 0) break; left = stk[--sp]; }
        }
    }
    w.t = w.best >= 0 ? tmax : 0.f;
}
void *rt_hostwalk_create(const float *verts, int n) {
    HostWalk *w = new HostWalk();
    w->n = n;
    w->r = rtbvh::build(verts, n);
    if (!w->r.ok) { delete w; return nullptr; }
    if (const char *e = rtbvh::knob("RT_BVH_WIDE")) w->wide = atoi(e) != 0;
    w->tris = leaf_order_triangles(verts, w->r, n);
    w->inverse.assign(n, 0);
    for (int k = 0; k < n; k++) w->inverse[w->r.order[k]] = k;
    w->ref.tree = rtref::build(verts, n);
    w->ref.root_leaf = w->ref.tree.nodes[0].count > 0 || n == 0;
    w->ref.leaf_of.assign((size_t)std::max(n, 1), 0);
    w->ref.parent.assign(w->ref.tree.nodes.size(), -1);
    for (size_t k = 0; k < w->ref.tree.nodes.size(); k++) {
        const rtref::Node &nd = w->ref.tree.nodes[k];
        for (int i = nd.link; nd.count > 0 && i < nd.link + nd.count; i++) w->ref.leaf_of[(size_t)w->ref.tree.prims[(size_t)i]] = (int)k;
        if (nd.count == 0 && n > 0) w->ref.parent[(size_t)nd.link] = w->ref.parent[(size_t)nd.link + 1] = (int)k;
    }
    return w;
}
void rt_hostwalk_destroy(void *h) { delete (HostWalk *)h; }
// work counters of the walks since the last reset: [0] rays [1] node steps [2] triangle tests [3] leaves visited
static long long g_walk_stats[4] = {0, 0, 0, 0};
static long long g_sp_hist[32] = {0};
// node steps of the walks since the last reset, by stack size at the step (what an LDS stack of a given depth would hold)
void rt_hostwalk_sp_hist(long long *out32, int reset) {
    for (int k = 0; k < 32; k++) {
        out32[k] = g_sp_hist[k];
        if (reset) g_sp_hist[k] = 0;
    }
}
void rt_hostwalk_stats(long long *out4, int reset) {
    for (int k = 0; k < 4; k++) {
        out4[k] = g_walk_stats[k];
        if (reset) g_walk_stats[k] = 0;
    }
}
// mode 0: closest hit -> out_i = original triangle index or -1, out_t = t;  mode 1: any hit excluding excluded[i]
// (original index or -1) -> out_i = 0 / 1.  Returns the number of rays whose walk failed (stack / step bound).
int rt_hostwalk_trace(void *h, int mode, int n_rays, const float *o3, const float *d3, const float *tmax_in, const int *excluded,
                      int *out_i, float *out_t) {
    const HostWalk &w = *(const HostWalk *)h;
    const rtbvh::Result &r = w.r;
    const std::vector<rtbvh::Pair> padded = w.wide ? padded_quads(r, n_rays, o3) : std::vector<rtbvh::Pair>();
    const std::vector<rtbvh::Pair> &rec = w.wide ? padded : r.pairs;
    const int stack_entries = w.wide ? r.stack_bound : r.pair_depth + 1;
    int failures = 0;
    long long st_nodes = 0, st_tris = 0, st_leaves = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : failures, st_nodes, st_tris, st_leaves)
    for (int i = 0; i < n_rays; i++) {
        V3 o{o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]}, d{d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]};
        const int excl = (mode == 1 && excluded[i] >= 0 && excluded[i] < w.n) ? w.inverse[excluded[i]] : -1;
        WalkResult res = walk_ray(rec, w.wide, w.tris, r.order, stack_entries, mode, o, d, tmax_in[i], excl);
        failures += res.failed ? 1 : 0;
        st_nodes += res.nodes; st_tris += res.tris; st_leaves += res.leaves;
#pragma omp critical
        for (int k = 0; k < 32; k++) g_sp_hist[k] += res.sp_hist[k];
        if (mode == 1) {
            out_i[i] = res.occluded ? 1 : 0;
        } else {
            out_i[i] = res.best >= 0 ? r.order[res.best] : -1;
            out_t[i] = res.t;
        }
    }
    g_walk_stats[0] += n_rays;
    g_walk_stats[1] += st_nodes;
    g_walk_stats[2] += st_tris;
    g_walk_stats[3] += st_leaves;
    return failures;
}
// The DEFAULT kernels' decision procedure on the CPU (k_paths / k_trace with VERIFY): the product's own walk, then
//   closest hit: the hit stands if the reference's walk can see its triangle (ref_visible) and no exact tie occurred at
//                the final distance; otherwise -- about one ray in 10^7 -- the ray is re-traced by the literal walk;
//   any hit:     the first accepted hit occludes if the reference's walk can see its triangle; if it cannot (~1 in 10^7),
//                that hit says nothing about the rest of the ray, and the literal walk decides;
//   (a ray with a -0.0 direction component: ref_visible tests every ancestor, see RefView).
// The result must equal the reference's literal walk on EVERY ray (tests/test_traversal_audit.py).
// stats6 += [rays, hits whose own box failed, hits whose leaf box failed (= hits the reference loses), exact ties at the
// final distance, -0.0 rays, literal re-traces]
int rt_hostwalk_trace_verified(void *h, int mode, int n_rays, const float *o3, const float *d3, const float *tmax_in,
                               const int *excluded, int *out_i, float *out_t, long long *stats6) {
    const HostWalk &w = *(const HostWalk *)h;
    const rtbvh::Result &r = w.r;
    const std::vector<rtbvh::Pair> padded = w.wide ? padded_quads(r, n_rays, o3) : std::vector<rtbvh::Pair>();
    const std::vector<rtbvh::Pair> &rec = w.wide ? padded : r.pairs;
    const int stack_entries = w.wide ? r.stack_bound : r.pair_depth + 1;
    int failures = 0;
    long long own = 0, leafb = 0, ties = 0, negz = 0, lit = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : failures, own, leafb, ties, negz, lit)
    for (int i = 0; i < n_rays; i++) {
        V3 o{o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]}, d{d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]};
        const int excl = (mode == 1 && excluded[i] >= 0 && excluded[i] < w.n) ? w.inverse[excluded[i]] : -1;
        const RefSlab s = ref_slab(o, d);
        WalkResult res;
        negz += s.neg_zero ? 1 : 0;
        if (mode == 1) {
            res = walk_ray(rec, w.wide, w.tris, r.order, stack_entries, 1, o, d, tmax_in[i], excl, &w.ref);
            if (res.unseen_occluder) {
                lit++;
                const bool failed = res.failed;
                const long long of = res.own_fail, lf = res.leaf_fail;
                literal_walk(w, 1, o, d, tmax_in[i], excl, res);
                res.failed = failed; res.own_fail = of; res.leaf_fail = lf;
            }
        } else {
            res = walk_ray(rec, w.wide, w.tris, r.order, stack_entries, 0, o, d, tmax_in[i], -1);
            if (res.best >= 0) {
                const bool tie = res.tie_t == res.t;
                const bool seen = ref_visible(w.ref, s, w.tris[(size_t)res.best], r.order[(size_t)res.best], res);
                ties += tie ? 1 : 0;
                if (tie || !seen) {
                    lit++;
                    const bool failed = res.failed;
                    literal_walk(w, 0, o, d, tmax_in[i], -1, res);
                    res.failed = failed;
                }
            }
        }
        failures += res.failed ? 1 : 0;
        own += res.own_fail; leafb += res.leaf_fail;
        if (mode == 1) {
            out_i[i] = res.occluded ? 1 : 0;
        } else {
            out_i[i] = res.best >= 0 ? r.order[res.best] : -1;
            out_t[i] = res.t;
        }
    }
    if (stats6) { stats6[0] += n_rays; stats6[1] += own; stats6[2] += leafb; stats6[3] += ties; stats6[4] += negz; stats6[5] += lit; }
    return failures;
}
// The reference's own tree as the product builds it for RT_FLAG_REFERENCE_WALK (rt_ref_tree.h), for a node-by-node
// comparison with the oracle's restatement of Bvh::Bvh (tests/test_host_logic.py).  Call with bounds6 == nullptr for the
// sizes: returns the node count; *depth receives the tree depth.  prims: reference position -> caller's triangle index.
int rt_ref_tree_export(const float *tri9, int n, float *bounds6, int *count, int *link, int *prims, int *depth) {
    const rtref::Tree t = rtref::build(tri9, n);
    if (depth) *depth = t.depth;
    if (bounds6) {
        for (size_t i = 0; i < t.nodes.size(); i++) {
            memcpy(bounds6 + 6 * i, t.nodes[i].box.b, 24);
            count[i] = t.nodes[i].count;
            link[i] = t.nodes[i].link;
        }
        for (size_t i = 0; i < t.prims.size(); i++) prims[i] = t.prims[i];
    }
    return (int)t.nodes.size();
}
}
