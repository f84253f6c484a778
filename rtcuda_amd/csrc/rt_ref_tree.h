// rt_ref_tree.h -- the REFERENCE's own acceleration structure, built by the product (host, header-only, C++17).
//
// Only RT_FLAG_REFERENCE_WALK uses it.  Two results of lashhw/rtcuda depend on the shape of its own tree and not on
// the scene alone: its fp32 slab test on exact boxes (aabb_intersector.cuh:14-36) drops about one accepted hit in
// 10^7 rays, and among hits at exactly equal t the triangle its walk tests LAST wins (triangle.cuh:49).  The default
// kernels define both by the triangle list (conservative box test, ties by caller index); this header exists so that
// the opt-in mode can walk exactly the tree the reference would have built and make exactly its decisions.
//
// What has to be equal to Bvh::Bvh (bvh.cuh:30-219), and is:
//   * per-triangle boxes and centres from the STORED triangle record {p0, e1 = p0 - p1, e2 = p2 - p0}: p1 and p2 are
//     recomputed as p0 - e1 and p0 + e2 (triangle.cuh:9-11,22-37), centre = (p0 + p1 + p2) * (1/3);
//   * three index arrays sorted per axis with std::sort and the comparator `centre[i] < centre[j]` (:74-86) -- an
//     unstable sort, so the order among equal keys is libstdc++'s introsort's; this file is compiled against the same
//     libstdc++ as the oracle, which is what makes the orders agree;
//   * full-sweep SAH: cost = half_area(left) * n_left + half_area(right) * n_right, axes 0, 1, 2 in that order, split
//     positions ascending, strict `<` (:126-141);
//   * a node stays a leaf if it has <= 1 triangle, sits at depth >= 30, or best_cost >= half_area(node) * (n - 1)
//     (:112,144-145);
//   * the other two axes' arrays are partitioned stably by side (:170-175);
//   * children are created together (left, then right, adjacent) and the SMALLER side is built first (:187-199) --
//     node numbers depend on that, the walk does not, but the numbering is kept so that the structure can be compared
//     with the oracle's node for node;
//   * triangles are stored in the order of the x-axis array (:208).
#ifndef RT_REF_TREE_H
#define RT_REF_TREE_H

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace rtref {

constexpr int kMaxDepth = 30;  // constant.hpp:7 BVH_MAX_DEPTH

// bounding_box.cuh:15 layout: [xmin, xmax, ymin, ymax, zmin, zmax]
struct Bounds {
    float b[6];
    void clear() {
        b[0] = b[2] = b[4] = FLT_MAX;
        b[1] = b[3] = b[5] = -FLT_MAX;
    }
    void grow(const Bounds &o) {
        for (int a = 0; a < 3; a++) {
            b[2 * a] = fminf(b[2 * a], o.b[2 * a]);
            b[2 * a + 1] = fmaxf(b[2 * a + 1], o.b[2 * a + 1]);
        }
    }
    float half_area() const {  // bounding_box.cuh:27-32: (ex + ey) * ez + ex * ey
        const float ex = b[1] - b[0], ey = b[3] - b[2], ez = b[5] - b[4];
        return (ex + ey) * ez + ex * ey;
    }
};

// One 32-byte node, as the walk reads it (bvh.cuh:5-14): count > 0 -> leaf over prims [link, link + count);
// count == 0 -> inner node whose children are nodes link and link + 1.
struct Node {
    Bounds box;
    int32_t count;
    int32_t link;
};
static_assert(sizeof(Node) == 32, "reference node is 32 bytes");

struct Tree {
    std::vector<Node> nodes;
    std::vector<int32_t> prims;  // position in the reference's primitive array -> the caller's triangle index
    int depth = 0;
};

// tri9: n x {p0, p1, p2}.  The file is built with -ffp-contract=off: every operation below is rounded on its own.
inline Tree build(const float *tri9, int n) {
    Tree t;
    t.nodes.resize(2 * (size_t)std::max(n, 1));
    t.nodes[0].box.clear();
    t.nodes[0].count = n;  // (n == 0: an empty leaf; the walk tests nothing)
    t.nodes[0].link = 0;
    std::vector<Bounds> box((size_t)n);
    std::vector<float> centre[3];
    for (auto &c : centre) c.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        const float *q = tri9 + 9 * (size_t)i;
        float p1[3], p2[3];
        for (int a = 0; a < 3; a++) {
            const float e1 = q[a] - q[3 + a], e2 = q[6 + a] - q[a];  // triangle.cuh:7
            p1[a] = q[a] - e1;                                      // :9
            p2[a] = q[a] + e2;                                      // :10
            box[i].b[2 * a] = fminf(q[a], fminf(p1[a], p2[a]));
            box[i].b[2 * a + 1] = fmaxf(q[a], fmaxf(p1[a], p2[a]));
            centre[a][i] = ((q[a] + p1[a]) + p2[a]) * (1.f / 3.f);  // :11
        }
        t.nodes[0].box.grow(box[i]);
    }
    std::vector<int32_t> ord[3];
    for (int a = 0; a < 3; a++) {
        ord[a].resize((size_t)n);
        std::iota(ord[a].begin(), ord[a].end(), 0);
        const float *key = centre[a].data();
        std::sort(ord[a].begin(), ord[a].end(), [key](int i, int j) { return key[i] < key[j]; });
    }
    std::vector<float> right_cost((size_t)n);
    std::vector<char> goes_left((size_t)n);
    int used = 1;
    // split [lo, hi) under node `self` at `level`; recursion depth <= kMaxDepth
    auto split = [&](auto &&again, int self, int lo, int hi, int level) -> void {
        const int count = hi - lo;
        t.nodes[self].count = count;  // leaf unless a split is accepted below
        t.nodes[self].link = lo;
        if (count <= 1 || level >= kMaxDepth) return;
        float best = FLT_MAX;
        int axis = -1, cut = -1;
        for (int a = 0; a < 3; a++) {
            const int32_t *r = ord[a].data();
            Bounds acc;
            acc.clear();
            for (int i = hi - 1; i > lo; i--) {
                acc.grow(box[r[i]]);
                right_cost[i] = acc.half_area() * (hi - i);
            }
            acc.clear();
            for (int i = lo; i < hi - 1; i++) {
                acc.grow(box[r[i]]);
                const float c = acc.half_area() * (i + 1 - lo) + right_cost[i + 1];
                if (c < best) {
                    best = c;
                    axis = a;
                    cut = i + 1;
                }
            }
        }
        if (best >= t.nodes[self].box.half_area() * (count - 1)) return;
        const int left = used, right = used + 1;
        used += 2;
        t.nodes[left].box.clear();
        t.nodes[right].box.clear();
        for (int i = lo; i < hi; i++) {
            const int k = ord[axis][i];
            const bool l = i < cut;
            t.nodes[l ? left : right].box.grow(box[k]);
            goes_left[k] = l ? 1 : 0;
        }
        for (int a = 0; a < 3; a++)
            if (a != axis)
                std::stable_partition(ord[a].begin() + lo, ord[a].begin() + hi, [&](int k) { return goes_left[k] != 0; });
        t.nodes[self].count = 0;
        t.nodes[self].link = left;
        t.depth = std::max(t.depth, level + 1);
        if (cut - lo < hi - cut) {  // the smaller side first
            again(again, left, lo, cut, level + 1);
            again(again, right, cut, hi, level + 1);
        } else {
            again(again, right, cut, hi, level + 1);
            again(again, left, lo, cut, level + 1);
        }
    };
    split(split, 0, 0, n, 0);
    t.nodes.resize((size_t)used);
    t.prims.assign(ord[0].begin(), ord[0].end());
    return t;
}

}  // namespace rtref
#endif  // RT_REF_TREE_H
