// rt_host_check.cpp -- host-only self-check of the BVH builder (rt_bvh.h), compiled with g++.
//
// Runs without a GPU: validates the 4-wide quantised tree the kernels will walk (every triangle in
// exactly one leaf, every node reachable exactly once, every DECODED child box contains its
// triangles, stack bound) and walks it on the CPU with the same control flow and the same fp32
// expressions as k_trace in rtcuda_amd.hip, comparing against an exhaustive search.  A malformed
// tree would hang or fault the GPU; this is where it is caught first.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "rt_bvh.h"

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
inline bool box_hit(V3 o, V3 inv, const float *lo, const float *hi, float tmax, float &entry) {
    float ax = (lo[0] - o.x) * inv.x, bx = (hi[0] - o.x) * inv.x;
    float ay = (lo[1] - o.y) * inv.y, by = (hi[1] - o.y) * inv.y;
    float az = (lo[2] - o.z) * inv.z, bz = (hi[2] - o.z) * inv.z;
    float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    entry = t_in;
    t_out = t_out * 1.000001f;  // (as box_hit in rtcuda_amd.hip; the kernels' 1 / d is v_rcp_f32, 1 ulp, here an exact division)
    return t_in <= t_out && t_out >= 0.f && t_in <= tmax * 1.000001f;
}
inline void child_box(const rtbvh::Node4 &nd, int k, float *lo, float *hi) {
    for (int a = 0; a < 3; a++) {
        float cell = rtbvh::cell_size(nd.exps, a);
        lo[a] = rtbvh::decode(nd.origin[a], (nd.qlo[a] >> (8 * k)) & 0xffu, cell);
        hi[a] = rtbvh::decode(nd.origin[a], (nd.qhi[a] >> (8 * k)) & 0xffu, cell);
    }
}
}  // namespace

extern "C" {
// out: [nodes, leaves, depth4, max_leaf_size, structural_errors, walk_mismatches, max_stack, max_steps,
//       stack_bound, binary_depth]
int rt_bvh_selfcheck(const float *verts, int n, int n_rays, const float *o3, const float *d3, int64_t *out10) {
    rtbvh::Result r = rtbvh::build(verts, n);
    memset(out10, 0, 10 * sizeof(int64_t));
    out10[0] = (int64_t)r.nodes.size();
    out10[1] = r.num_leaves;
    out10[2] = r.max_depth;
    out10[8] = r.stack_bound;
    out10[9] = r.bin_depth;
    int64_t errors = r.ok ? 0 : 1;
    std::vector<int> seen_tri(n, 0), seen_node(r.nodes.size(), 0);
    seen_node[0] = 1;
    int max_leaf = 0;
    // subtree boxes (from decoded children) must nest: collect each node's decoded union on the way
    for (size_t ni = 0; ni < r.nodes.size(); ni++) {
        const rtbvh::Node4 &nd = r.nodes[ni];
        for (int k = 0; k < 4; k++) {
            int link = nd.link[k];
            if (link == rtbvh::kNoChild) continue;
            float lo[3], hi[3];
            child_box(nd, k, lo, hi);
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
                            if (!(v[3 * c + a] > lo[a] || v[3 * c + a] == lo[a]) || !(v[3 * c + a] < hi[a] || v[3 * c + a] == hi[a])) errors++;
                }
            } else {
                if (link >= (int)r.nodes.size() || link <= (int)ni) { errors++; continue; }  // children come later
                seen_node[link]++;
                // every decoded grandchild box must lie inside this decoded child box
                const rtbvh::Node4 &ch = r.nodes[link];
                for (int g = 0; g < 4; g++) {
                    if (ch.link[g] == rtbvh::kNoChild) continue;
                    float glo[3], ghi[3];
                    child_box(ch, g, glo, ghi);
                    for (int a = 0; a < 3; a++) {
                        float tol = 1e-2f * (hi[a] - lo[a]) + 1e-6f;  // the child's own grid is finer than the parent's cell
                        if (glo[a] < lo[a] - tol || ghi[a] > hi[a] + tol) errors++;
                    }
                }
            }
        }
    }
    for (int i = 0; i < n; i++) if (seen_tri[i] != 1) errors++;
    for (size_t i = 0; i < r.nodes.size(); i++) if (seen_node[i] != 1) errors++;
    out10[3] = max_leaf;
    out10[4] = errors;
    if (errors) return 0;
    // CPU walk with the kernel's control flow vs exhaustive search
    std::vector<Tri> tris(n);
    for (int k = 0; k < n; k++) {
        const float *q = verts + 9 * (size_t)r.order[k];
        V3 p0{q[0], q[1], q[2]}, p1{q[3], q[4], q[5]}, p2{q[6], q[7], q[8]};
        tris[k].p0 = p0; tris[k].e1 = sub(p0, p1); tris[k].e2 = sub(p2, p0); tris[k].n = cross(tris[k].e1, tris[k].e2);
    }
    int64_t mism = 0, max_stack = 0, max_steps = 0, sum_inner = 0, sum_leaf = 0, sum_tri = 0;
    std::vector<int> stack(r.stack_bound + 8);
    for (int i = 0; i < n_rays; i++) {
        V3 o{o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]}, d{d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]};
        auto clampinv = [](float x) { return 1.f / ((fabsf(x) < FLT_EPSILON) ? copysignf(FLT_EPSILON, x) : x); };
        V3 inv{clampinv(d.x), clampinv(d.y), clampinv(d.z)};
        float tmax = FLT_MAX, t;
        int best = -1, sp = 0, cur = 0;
        int64_t steps = 0;
        bool bad = false;
        while (cur != rtbvh::kNoChild) {
            if (++steps > 1000000) { bad = true; break; }
            if (cur >= 0) {
                sum_inner++;
                const rtbvh::Node4 &nd = r.nodes[cur];
                uint32_t key[4];
                int lnk[4];
                for (int k = 0; k < 4; k++) {
                    float lo[3], hi[3], e;
                    child_box(nd, k, lo, hi);
                    bool h = box_hit(o, inv, lo, hi, tmax, e) && nd.link[k] != rtbvh::kNoChild;
                    float ee = fmaxf(e, 0.f);
                    uint32_t bits;
                    memcpy(&bits, &ee, 4);
                    key[k] = h ? bits : 0xffffffffu;
                    lnk[k] = nd.link[k];
                }
                auto cswap = [&](int a, int b) { if (key[b] < key[a]) { std::swap(key[a], key[b]); std::swap(lnk[a], lnk[b]); } };
                cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
                for (int k = 3; k >= 1; k--)
                    if (key[k] != 0xffffffffu) {
                        if (sp >= (int)stack.size()) { bad = true; break; }
                        stack[sp++] = lnk[k];
                    }
                if (bad) break;
                if (sp > max_stack) max_stack = sp;
                if (key[0] != 0xffffffffu) cur = lnk[0];
                else cur = sp > 0 ? stack[--sp] : rtbvh::kNoChild;
            } else {
                int ref = ~cur, first = ref >> 3, count = ref & 7;
                sum_leaf++;
                sum_tri += count;
                for (int k = first; k < first + count; k++) if (tri_hit(tris[k], o, d, tmax, t)) { tmax = t; best = k; }
                cur = sp > 0 ? stack[--sp] : rtbvh::kNoChild;
            }
        }
        if (bad) { mism += 1000000; continue; }
        if (steps > max_steps) max_steps = steps;
        float bt = FLT_MAX; int bb = -1;
        for (int k = 0; k < n; k++) if (tri_hit(tris[k], o, d, bt, t)) { bt = t; bb = k; }
        if ((bb < 0) != (best < 0) || (bb >= 0 && bt != tmax)) mism++;
    }
    // ---- the 2-wide records of the same tree: same walk, two exact boxes per record
    {
        int64_t pair_inner = 0;
        std::vector<int> pseen(r.pairs.size(), 0);
        pseen[0] = 1;
        for (size_t pi = 0; pi < r.pairs.size(); pi++)
            for (int side = 0; side < 2; side++) {
                int link = side ? r.pairs[pi].rlink : r.pairs[pi].llink;
                if (link >= 0) { if (link >= (int)r.pairs.size() || link <= (int)pi) mism += 1000000; else pseen[link]++; }
            }
        for (size_t i = 0; i < r.pairs.size(); i++) if (pseen[i] != 1) mism += 1000000;
        std::vector<int> pstack(r.pair_depth + 8);
        for (int i = 0; i < n_rays && mism < 1000000; i++) {
            V3 o{o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]}, d{d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]};
            auto clampinv = [](float x) { return 1.f / ((fabsf(x) < FLT_EPSILON) ? copysignf(FLT_EPSILON, x) : x); };
            V3 inv{clampinv(d.x), clampinv(d.y), clampinv(d.z)};
            float tmax = FLT_MAX, t;
            int best = -1, sp = 0, cur = 0;
            int64_t steps = 0;
            while (cur != rtbvh::kNoChild) {
                if (++steps > 1000000) { mism += 1000000; break; }
                if (cur >= 0) {
                    pair_inner++;
                    const rtbvh::Pair &p = r.pairs[cur];
                    float el, er;
                    bool hl = box_hit(o, inv, p.lbox, p.lbox + 3, tmax, el) && p.llink != rtbvh::kNoChild;
                    bool hr = box_hit(o, inv, p.rbox, p.rbox + 3, tmax, er) && p.rlink != rtbvh::kNoChild;
                    if (hl && hr) {
                        bool lf = !(el > er);
                        if (sp >= (int)pstack.size()) { mism += 1000000; break; }
                        pstack[sp++] = lf ? p.rlink : p.llink;
                        cur = lf ? p.llink : p.rlink;
                    } else if (hl) cur = p.llink;
                    else if (hr) cur = p.rlink;
                    else cur = sp > 0 ? pstack[--sp] : rtbvh::kNoChild;
                } else {
                    int ref = ~cur, first = ref >> 3, count = ref & 7;
                    for (int k = first; k < first + count; k++) if (tri_hit(tris[k], o, d, tmax, t)) { tmax = t; best = k; }
                    cur = sp > 0 ? pstack[--sp] : rtbvh::kNoChild;
                }
            }
            float bt = FLT_MAX; int bb = -1;
            for (int k = 0; k < n; k++) if (tri_hit(tris[k], o, d, bt, t)) { bt = t; bb = k; }
            if ((bb < 0) != (best < 0) || (bb >= 0 && bt != tmax)) mism++;
        }
        if (getenv("RT_BVH_STATS") && n_rays > 0) fprintf(stderr, "pair walk: inner %.2f per ray, %zu pairs, depth %d\n", (double)pair_inner / n_rays, r.pairs.size(), r.pair_depth);
    }
    out10[5] = mism; out10[6] = max_stack; out10[7] = max_steps;
    if (getenv("RT_BVH_STATS") && n_rays > 0)
        fprintf(stderr, "bvh walk: inner %.2f leaf %.2f tri %.2f per ray\n", (double)sum_inner / n_rays, (double)sum_leaf / n_rays, (double)sum_tri / n_rays);
    return 0;
}

// ---- CPU walk of the product's default tree (2-wide records, padded boxes) for arbitrary rays: the control flow and
// the fp32 expressions of inner_step<false> + the triangle blocks of k_trace / k_paths in rtcuda_amd.hip.  Used by the
// traversal audit (tests/test_traversal_audit.py): rays logged from an oracle render are replayed here, on the
// oracle's reference traversal and through exhaustive search, to show which of the two BVH walks loses hits.
struct HostWalk {
    rtbvh::Result r;
    std::vector<Tri> tris;     // leaf order
    std::vector<int> inverse;  // original index -> leaf-order index
    int n = 0;
};
void *rt_hostwalk_create(const float *verts, int n) {
    HostWalk *w = new HostWalk();
    w->n = n;
    w->r = rtbvh::build(verts, n);
    if (!w->r.ok) { delete w; return nullptr; }
    w->tris.resize(n);
    w->inverse.assign(n, 0);
    for (int k = 0; k < n; k++) {
        const float *q = verts + 9 * (size_t)w->r.order[k];
        V3 p0{q[0], q[1], q[2]}, p1{q[3], q[4], q[5]}, p2{q[6], q[7], q[8]};
        w->tris[k].p0 = p0; w->tris[k].e1 = sub(p0, p1); w->tris[k].e2 = sub(p2, p0); w->tris[k].n = cross(w->tris[k].e1, w->tris[k].e2);
        w->inverse[w->r.order[k]] = k;
    }
    return w;
}
void rt_hostwalk_destroy(void *h) { delete (HostWalk *)h; }
// mode 0: closest hit -> out_i = original triangle index or -1, out_t = t;  mode 1: any hit excluding excluded[i]
// (original index or -1) -> out_i = 0 / 1
// work counters of the walks since the last reset: [0] rays [1] node steps [2] triangle tests [3] leaves visited
static long long g_walk_stats[4] = {0, 0, 0, 0};
void rt_hostwalk_stats(long long *out4, int reset) {
    for (int k = 0; k < 4; k++) {
        out4[k] = g_walk_stats[k];
        if (reset) g_walk_stats[k] = 0;
    }
}
int rt_hostwalk_trace(void *h, int mode, int n_rays, const float *o3, const float *d3, const float *tmax_in, const int *excluded,
                      int *out_i, float *out_t) {
    const HostWalk &w = *(const HostWalk *)h;
    const rtbvh::Result &r = w.r;
    int failures = 0;
    long long st_nodes = 0, st_tris = 0, st_leaves = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : failures, st_nodes, st_tris, st_leaves)
    for (int i = 0; i < n_rays; i++) {
        std::vector<int> pstack(r.pair_depth + 8);
        V3 o{o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]}, d{d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]};
        auto clampinv = [](float x) { return 1.f / ((fabsf(x) < FLT_EPSILON) ? copysignf(FLT_EPSILON, x) : x); };
        V3 inv{clampinv(d.x), clampinv(d.y), clampinv(d.z)};
        float tmax = tmax_in[i], t;
        const int excl = (mode == 1 && excluded[i] >= 0 && excluded[i] < w.n) ? w.inverse[excluded[i]] : -1;
        int best = -1, sp = 0, cur = w.n > 0 ? 0 : rtbvh::kNoChild;
        bool occluded = false;
        int64_t steps = 0;
        while (cur != rtbvh::kNoChild && !occluded) {
            if (++steps > 1000000) { failures++; break; }
            if (cur >= 0) {
                st_nodes++;
                const rtbvh::Pair &p = r.pairs[cur];
                float el, er;
                bool hl = box_hit(o, inv, p.lbox, p.lbox + 3, tmax, el) && p.llink != rtbvh::kNoChild;
                bool hr = box_hit(o, inv, p.rbox, p.rbox + 3, tmax, er) && p.rlink != rtbvh::kNoChild;
                if (hl && hr) {
                    bool lf = !(el > er);
                    if (sp >= (int)pstack.size()) { failures++; break; }
                    pstack[sp++] = lf ? p.rlink : p.llink;
                    cur = lf ? p.llink : p.rlink;
                } else if (hl) cur = p.llink;
                else if (hr) cur = p.rlink;
                else cur = sp > 0 ? pstack[--sp] : rtbvh::kNoChild;
            } else {
                int ref = ~cur, first = ref >> 3, count = ref & 7;
                st_leaves++;
                for (int k = first; k < first + count; k++) {
                    st_tris++;
                    if (tri_hit(w.tris[k], o, d, tmax, t)) {
                        if (mode == 1) {
                            if (k != excl) { occluded = true; break; }
                        } else if (!(t == tmax && best >= 0) || r.order[k] > r.order[best]) {  // closest_hit_wins()
                            tmax = t;
                            best = k;
                        }
                    }
                }
                cur = sp > 0 ? pstack[--sp] : rtbvh::kNoChild;
            }
        }
        if (mode == 1) {
            out_i[i] = occluded ? 1 : 0;
        } else {
            out_i[i] = best >= 0 ? r.order[best] : -1;
            out_t[i] = best >= 0 ? tmax : 0.f;
        }
    }
    g_walk_stats[0] += n_rays;
    g_walk_stats[1] += st_nodes;
    g_walk_stats[2] += st_tris;
    g_walk_stats[3] += st_leaves;
    return failures;
}
}
