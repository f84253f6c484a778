// rt_host_check.cpp -- host-only self-check of the BVH builder (rt_bvh.h), compiled with g++.
//
// Runs without a GPU: validates the pair-record tree the kernels will walk (every triangle in
// exactly one leaf, every pair reachable exactly once, child boxes contain their triangles, depth
// within the kernel's LDS stack) and walks it on the CPU with the same control flow as
// traverse_closest in rtcuda_amd.hip, comparing against an exhaustive search.  A malformed tree
// would hang or fault the GPU; this is where it is caught first.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
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
inline bool box_hit(V3 o, V3 inv, const float *b, float tmax, float &entry) {
    float ax = (b[0] - o.x) * inv.x, bx = (b[3] - o.x) * inv.x;
    float ay = (b[1] - o.y) * inv.y, by = (b[4] - o.y) * inv.y;
    float az = (b[2] - o.z) * inv.z, bz = (b[5] - o.z) * inv.z;
    float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    entry = t_in;
    t_out = t_out * 1.0000004f;
    return t_in <= t_out && t_out >= 0.f && t_in <= tmax;
}
}  // namespace

extern "C" {
// out: [pairs, leaves, max_depth, max_leaf_size, structural_errors, walk_mismatches, max_stack, max_steps]
int rt_bvh_selfcheck(const float *verts, int n, int n_rays, const float *o3, const float *d3, int64_t *out8) {
    rtbvh::Result r = rtbvh::build(verts, n);
    memset(out8, 0, 8 * sizeof(int64_t));
    out8[0] = (int64_t)r.pairs.size();
    out8[1] = r.num_leaves;
    out8[2] = r.max_depth;
    int64_t errors = 0;
    std::vector<int> seen_tri(n, 0), seen_pair(r.pairs.size(), 0);
    seen_pair[0] = 1;
    int max_leaf = 0;
    for (size_t pi = 0; pi < r.pairs.size(); pi++) {
        const rtbvh::Pair &p = r.pairs[pi];
        for (int side = 0; side < 2; side++) {
            int link = side ? p.rlink : p.llink, count = side ? p.rcount : p.lcount;
            const float *box = side ? p.rbox : p.lbox;
            if (count > 0) {
                if (link < 0 || link + count > n) { errors++; continue; }
                max_leaf = std::max(max_leaf, count);
                for (int k = link; k < link + count; k++) {
                    int ti = r.order[k];
                    if (ti < 0 || ti >= n) { errors++; continue; }
                    seen_tri[ti]++;
                    const float *v = verts + 9 * (size_t)ti;
                    for (int c = 0; c < 3; c++)
                        for (int a = 0; a < 3; a++)
                            if (v[3 * c + a] < box[a] || v[3 * c + a] > box[3 + a]) errors++;
                }
            } else if (link >= 0) {
                if (link >= (int)r.pairs.size() || link <= (int)pi) { errors++; continue; }  // children come later
                seen_pair[link]++;
                const rtbvh::Pair &ch = r.pairs[link];
                for (int a = 0; a < 3; a++) {  // child boxes inside the parent's box for that side (up to padding)
                    float lo = std::min(ch.lbox[a], ch.rbox[a]), hi = std::max(ch.lbox[3 + a], ch.rbox[3 + a]);
                    if (ch.llink == -1 && ch.lcount == 0) { lo = ch.rbox[a]; hi = ch.rbox[3 + a]; }
                    if (ch.rlink == -1 && ch.rcount == 0) { lo = ch.lbox[a]; hi = ch.lbox[3 + a]; }
                    float tol = 1e-5f * std::max(1.f, std::max(fabsf(lo), fabsf(hi)));
                    if (lo < box[a] - tol || hi > box[3 + a] + tol) errors++;
                }
            }
        }
    }
    for (int i = 0; i < n; i++) if (seen_tri[i] != 1) errors++;
    for (size_t i = 0; i < r.pairs.size(); i++) if (seen_pair[i] != 1) errors++;
    out8[3] = max_leaf;
    out8[4] = errors;
    if (errors) return 0;
    // CPU walk with the kernel's control flow vs exhaustive search
    std::vector<Tri> tris(n);
    for (int k = 0; k < n; k++) {
        const float *q = verts + 9 * (size_t)r.order[k];
        V3 p0{q[0], q[1], q[2]}, p1{q[3], q[4], q[5]}, p2{q[6], q[7], q[8]};
        tris[k].p0 = p0; tris[k].e1 = sub(p0, p1); tris[k].e2 = sub(p2, p0); tris[k].n = cross(tris[k].e1, tris[k].e2);
    }
    int64_t mism = 0, max_stack = 0, max_steps = 0;
    for (int i = 0; i < n_rays; i++) {
        V3 o{o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]}, d{d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]};
        auto clampinv = [](float x) { return 1.f / ((fabsf(x) < FLT_EPSILON) ? copysignf(FLT_EPSILON, x) : x); };
        V3 inv{clampinv(d.x), clampinv(d.y), clampinv(d.z)};
        float tmax = FLT_MAX, t;
        int best = -1, sp = 0, node = 0, stack[64];
        int64_t steps = 0;
        while (true) {
            if (++steps > 1000000) { mism += 1000000; break; }
            const rtbvh::Pair &p = r.pairs[node];
            float el, er;
            bool hl = box_hit(o, inv, p.lbox, tmax, el), hr = box_hit(o, inv, p.rbox, tmax, er);
            if (hl && p.lcount > 0) for (int k = p.llink; k < p.llink + p.lcount; k++) if (tri_hit(tris[k], o, d, tmax, t)) { tmax = t; best = k; }
            if (hr && p.rcount > 0) for (int k = p.rlink; k < p.rlink + p.rcount; k++) if (tri_hit(tris[k], o, d, tmax, t)) { tmax = t; best = k; }
            bool il = hl && p.lcount == 0 && p.llink >= 0, ir = hr && p.rcount == 0 && p.rlink >= 0;
            if (il && ir) {
                int nr = el > er ? p.rlink : p.llink, fr = el > er ? p.llink : p.rlink;
                if (sp >= 64) { mism += 1000000; break; }
                stack[sp++] = fr; node = nr;
                if (sp > max_stack) max_stack = sp;
            } else if (il) node = p.llink;
            else if (ir) node = p.rlink;
            else { if (sp == 0) break; node = stack[--sp]; }
        }
        if (steps > max_steps) max_steps = steps;
        float bt = FLT_MAX; int bb = -1;
        for (int k = 0; k < n; k++) if (tri_hit(tris[k], o, d, bt, t)) { bt = t; bb = k; }
        if ((bb < 0) != (best < 0) || (bb >= 0 && bt != tmax)) mism++;
    }
    out8[5] = mism; out8[6] = max_stack; out8[7] = max_steps;
    return 0;
}
}
