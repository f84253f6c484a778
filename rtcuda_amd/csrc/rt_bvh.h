// rt_bvh.h -- host-side BVH construction for the MI355X render path (header-only, C++17).
//
// Role in the reference: Bvh::Bvh (bvh.cuh:30-219) builds a binary SAH BVH on the host and uploads
// 32-byte nodes whose two children are adjacent.  This builder is NOT a restatement of it: the image
// only needs "closest accepted triangle" (results are topology-independent
// except equal-t ties), so the tree is built for the traversal kernel instead:
//
//   * full-sweep SAH over three index arrays kept sorted per axis (the quality class of the
//     reference's builder), leaf cost model C_leaf = n_tris, C_split = 1 + SAH, max 4 tris/leaf;
//   * output is an array of 64-byte PAIR records -- both children's boxes plus both child links in
//     one 64-byte line -- so one traversal step is one aligned 64-byte fetch (4 x dwordx4);
//   * triangles are re-ordered so every leaf's triangles are contiguous ("leaf order");
//   * boxes are padded outward by a few ulps so the fp32 slab test in the kernel is conservative
//     with respect to the exact triangle test (a box test must never cull a triangle the
//     triangle test would accept).
//
// Pair record layout (16 words):
//   w0..w5   left  box  xmin ymin zmin xmax ymax zmax
//   w6..w11  right box  xmin ymin zmin xmax ymax zmax
//   w12 left link, w13 right link, w14 left count, w15 right count
//   link >= 0 with count == 0 : index of the child's pair record
//   count  > 0                : leaf, link = first triangle (leaf order), count triangles
//   count == 0 and link == -1 : empty child (box inverted, never hit)
#ifndef RT_BVH_H
#define RT_BVH_H

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

namespace rtbvh {

struct Box {
    float lo[3], hi[3];
    void reset() {
        lo[0] = lo[1] = lo[2] = FLT_MAX;
        hi[0] = hi[1] = hi[2] = -FLT_MAX;
    }
    void extend(const Box &o) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], o.lo[a]);
            hi[a] = std::max(hi[a], o.hi[a]);
        }
    }
    float half_area() const {
        float e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
        return (e0 + e1) * e2 + e0 * e1;
    }
};

struct Pair {
    float lbox[6];
    float rbox[6];
    int32_t llink, rlink, lcount, rcount;
};
static_assert(sizeof(Pair) == 64, "pair record must be one 64-byte line");

struct Result {
    std::vector<Pair> pairs;
    std::vector<int32_t> order;  // leaf order -> original triangle index
    int max_depth = 0;
    int num_leaves = 0;
};

constexpr int kMaxLeaf = 4;
constexpr int kMaxDepth = 48;  // hard cap; the traversal stack is sized from Result::max_depth

// nextafter-style outward padding: k ulps at the magnitude of the value (min 1e-30 absolute)
inline float pad_down(float v, int k) {
    for (int i = 0; i < k; i++) v = std::nextafter(v, -FLT_MAX);
    return v;
}
inline float pad_up(float v, int k) {
    for (int i = 0; i < k; i++) v = std::nextafter(v, FLT_MAX);
    return v;
}

// verts: n x 9 floats (p0 p1 p2).  Deterministic for a given input.
inline Result build(const float *verts, int n) {
    Result res;
    res.order.resize(n);
    if (n == 0) {
        Pair p;
        Box e;
        e.reset();
        memcpy(p.lbox, e.lo, 12);
        memcpy(p.lbox + 3, e.hi, 12);
        memcpy(p.rbox, e.lo, 12);
        memcpy(p.rbox + 3, e.hi, 12);
        p.llink = p.rlink = -1;
        p.lcount = p.rcount = 0;
        res.pairs.push_back(p);
        return res;
    }
    std::vector<Box> boxes(n);
    std::vector<float> cen(3 * (size_t)n);
    for (int i = 0; i < n; i++) {
        const float *v = verts + 9 * (size_t)i;
        Box b;
        for (int a = 0; a < 3; a++) {
            b.lo[a] = std::min(v[a], std::min(v[3 + a], v[6 + a]));
            b.hi[a] = std::max(v[a], std::max(v[3 + a], v[6 + a]));
            cen[3 * (size_t)i + a] = 0.5f * (b.lo[a] + b.hi[a]);
        }
        boxes[i] = b;
    }
    std::vector<int32_t> idx[3];
    for (int a = 0; a < 3; a++) {
        idx[a].resize(n);
        std::iota(idx[a].begin(), idx[a].end(), 0);
        std::stable_sort(idx[a].begin(), idx[a].end(), [&](int i, int j) {
            return cen[3 * (size_t)i + a] < cen[3 * (size_t)j + a];
        });
    }
    std::vector<float> right_area(n);
    std::vector<uint8_t> side(n);
    std::vector<int32_t> tmp(n);

    struct Task {
        int begin, end, depth;
        int parent;   // pair index that owns this child (-1 for the root range)
        int which;    // 0 = left child of parent, 1 = right
        Box box;
    };
    auto set_child_box = [&](Pair &p, int which, const Box &b) {
        float *dst = which ? p.rbox : p.lbox;
        for (int a = 0; a < 3; a++) {
            dst[a] = pad_down(b.lo[a], 2);
            dst[3 + a] = pad_up(b.hi[a], 2);
        }
    };
    int out_pos = 0;  // next free position in leaf order
    auto make_leaf = [&](const Task &t) {
        Pair &p = res.pairs[t.parent];
        int first = out_pos;
        for (int i = t.begin; i < t.end; i++) res.order[out_pos++] = idx[0][i];
        if (t.which) {
            p.rlink = first;
            p.rcount = t.end - t.begin;
        } else {
            p.llink = first;
            p.lcount = t.end - t.begin;
        }
        res.num_leaves++;
        res.max_depth = std::max(res.max_depth, t.depth);
    };

    Box root;
    root.reset();
    for (int i = 0; i < n; i++) root.extend(boxes[i]);

    // The root range always becomes pair 0 (a scene that is a single leaf gets a pair whose right
    // child is empty), so the kernel never special-cases "root is a leaf" (bvh.cuh:252,307).
    std::vector<Task> stack;
    {
        Pair p;
        memset(&p, 0, sizeof(p));
        Box e;
        e.reset();
        memcpy(p.lbox, e.lo, 12);
        memcpy(p.lbox + 3, e.hi, 12);
        memcpy(p.rbox, e.lo, 12);
        memcpy(p.rbox + 3, e.hi, 12);
        p.llink = p.rlink = -1;
        res.pairs.push_back(p);
    }
    // split a range into two child tasks of pair `pi`; returns false if it should be a leaf
    auto try_split = [&](int begin, int end, const Box &box, int &best_axis, int &best_split, Box &lb, Box &rb) {
        int cnt = end - begin;
        float best = FLT_MAX;
        best_axis = -1;
        for (int a = 0; a < 3; a++) {
            Box acc;
            acc.reset();
            for (int i = end - 1; i > begin; i--) {
                acc.extend(boxes[idx[a][i]]);
                right_area[i] = acc.half_area();
            }
            acc.reset();
            for (int i = begin; i < end - 1; i++) {
                acc.extend(boxes[idx[a][i]]);
                float cost = acc.half_area() * (float)(i + 1 - begin) + right_area[i + 1] * (float)(end - i - 1);
                if (cost < best) {
                    best = cost;
                    best_axis = a;
                    best_split = i + 1;
                }
            }
        }
        if (best_axis < 0) return false;
        float leaf_cost = box.half_area() * (float)cnt;
        float split_cost = box.half_area() * 1.0f + best;  // traversal step ~ one triangle test
        if (cnt <= kMaxLeaf && split_cost >= leaf_cost) return false;
        lb.reset();
        rb.reset();
        for (int i = begin; i < best_split; i++) lb.extend(boxes[idx[best_axis][i]]);
        for (int i = best_split; i < end; i++) rb.extend(boxes[idx[best_axis][i]]);
        return true;
    };
    auto partition_other_axes = [&](int begin, int end, int axis, int split) {
        for (int i = begin; i < split; i++) side[idx[axis][i]] = 0;
        for (int i = split; i < end; i++) side[idx[axis][i]] = 1;
        for (int a = 0; a < 3; a++) {
            if (a == axis) continue;
            int l = begin, r = 0;
            for (int i = begin; i < end; i++) {
                int t = idx[a][i];
                if (side[t] == 0) idx[a][l++] = t;
                else tmp[r++] = t;
            }
            memcpy(&idx[a][l], tmp.data(), sizeof(int32_t) * (size_t)r);
        }
    };

    // root handling
    {
        int axis, split;
        Box lb, rb;
        if (n >= 2 && try_split(0, n, root, axis, split, lb, rb)) {
            partition_other_axes(0, n, axis, split);
            set_child_box(res.pairs[0], 0, lb);
            set_child_box(res.pairs[0], 1, rb);
            // depth-first, left child first: keeps subtrees contiguous in memory
            stack.push_back(Task{split, n, 1, 0, 1, rb});
            stack.push_back(Task{0, split, 1, 0, 0, lb});
        } else {
            set_child_box(res.pairs[0], 0, root);
            Task t{0, n, 1, 0, 0, root};
            make_leaf(t);
        }
    }
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        int cnt = t.end - t.begin;
        int axis = -1, split = -1;
        Box lb, rb;
        bool can_split = cnt >= 2 && t.depth < kMaxDepth && try_split(t.begin, t.end, t.box, axis, split, lb, rb);
        if (!can_split && cnt > kMaxLeaf && t.depth < kMaxDepth) {
            // SAH found no useful plane (coincident boxes): split in the middle of axis 0
            axis = 0;
            split = t.begin + cnt / 2;
            lb.reset();
            rb.reset();
            for (int i = t.begin; i < split; i++) lb.extend(boxes[idx[0][i]]);
            for (int i = split; i < t.end; i++) rb.extend(boxes[idx[0][i]]);
            can_split = true;
        }
        if (!can_split) {
            make_leaf(t);
            continue;
        }
        partition_other_axes(t.begin, t.end, axis, split);
        int pi = (int)res.pairs.size();
        Pair p;
        memset(&p, 0, sizeof(p));
        p.llink = p.rlink = -1;
        res.pairs.push_back(p);
        set_child_box(res.pairs[pi], 0, lb);
        set_child_box(res.pairs[pi], 1, rb);
        Pair &par = res.pairs[t.parent];
        if (t.which) {
            par.rlink = pi;
            par.rcount = 0;
        } else {
            par.llink = pi;
            par.lcount = 0;
        }
        stack.push_back(Task{split, t.end, t.depth + 1, pi, 1, rb});
        stack.push_back(Task{t.begin, split, t.depth + 1, pi, 0, lb});
    }
    return res;
}

}  // namespace rtbvh
#endif  // RT_BVH_H
