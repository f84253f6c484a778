// rt_bvh.h -- host-side BVH construction for the MI355X render path (header-only, C++17).
//
// Role in the reference: Bvh::Bvh (bvh.cuh:30-219) builds a binary SAH BVH on the host and uploads
// 32-byte nodes whose two children are adjacent.  This builder is NOT a restatement of it: the image
// only needs "closest accepted triangle" (results are topology-independent
// except equal-t ties), so the tree is built for the traversal kernel instead:
//
//   1. binary tree by full-sweep SAH over three index arrays kept sorted per axis (the quality class
//      of the reference's builder), leaf cost model C_leaf = n_tris, C_split = 1 + SAH, <= 4 tris/leaf;
//   1b. insertion-based optimisation of that tree (optimize_reinsert): -10 % node steps per ray;
//   2. written out in two formats made of the same 64-byte record -- two children's exact fp32 boxes, padded
//      outward by 2 ulps so that the kernels' slab test stays conservative with respect to the exact triangle
//      test (a box test must never cull a triangle the triangle test would accept), and their links
//      (>= 0 inner record index, < 0 leaf reference ~(first << 3 | count), 0x80000000 = no child):
//        `pairs`  the binary tree, one record per inner node;
//        `quads`  the tree collapsed to 4-wide (the child with the largest surface area is opened until a node
//                 has four children), TWO consecutive records per node (children 0, 1 | children 2, 3): half the
//                 dependent node fetches per ray for the same box arithmetic -- the default of the kernels.
//      (Round 1's 4-wide format squeezed a node into ONE record with 8-bit quantised boxes; decoding them cost
//      more vector instructions than the halved fetches saved: 1.8 vs 3.0 Gsamples/s.  Removed.)
//   3. triangles are re-ordered so every leaf's triangles are contiguous ("leaf order").
#ifndef RT_BVH_H
#define RT_BVH_H

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace rtbvh {

// Experiment knobs (RT_PERSISTENT, RT_MAJORITY, RT_BVH_WIDE, RT_BVH_MAX_LEAF, ...: DESIGN.md section 7) are read only when
// the process sets RTCUDA_EXPERIMENTAL=1: a drop-in library must not change behaviour because some RT_* name happens to be
// set in a user's shell.  Tests and tools set the gate; bench.py never does.
inline const char *knob(const char *name) {
    const char *gate = getenv("RTCUDA_EXPERIMENTAL");
    return (gate && gate[0] == '1' && gate[1] == 0) ? getenv(name) : nullptr;
}


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

constexpr int32_t kNoChild = (int32_t)0x80000000;
inline int32_t leaf_ref(int first, int count) { return ~((first << 3) | count); }

// A node of the collapsed 4-wide tree before it is written out as two pair-style records (see `quads`).
struct WideNode {
    int32_t link[4];  // child k: a node index (>= 0), a leaf reference (< 0) or kNoChild
    Box box[4];       // exact (unpadded) child boxes
};

// The 64-byte record both tree formats are made of: two children's exact fp32 boxes (padded outward by 2 ulps) and
// their links.  w0..w5 left box lo xyz / hi xyz, w6..w11 right box, w12 left link, w13 right link, w14, w15 spare.
// 2-wide tree: one record per inner node (`pairs`).  4-wide tree: two consecutive records per node (`quads`).
struct Pair {
    float lbox[6];
    float rbox[6];
    int32_t llink, rlink;
    int32_t spare[2];
};
static_assert(sizeof(Pair) == 64, "pair record must be one 64-byte line");
// A record of the 4-wide format without children: links kNoChild and all-+inf boxes, which no ray enters (per axis the
// slab distances of such a box are both +inf or both -inf), so the kernel's box test needs no look at the link.
inline Pair absent_quad_record() {
    Pair p;
    memset(&p, 0, sizeof(p));
    for (int a = 0; a < 6; a++) p.lbox[a] = p.rbox[a] = INFINITY;
    p.llink = p.rlink = (int32_t)0x80000000;
    return p;
}

struct BinNode {
    Box box;         // exact (unpadded) box of the subtree
    int left = -1;   // children (binary) or -1
    int right = -1;
    int first = 0;   // leaf: first triangle in leaf order
    int count = 0;   // leaf: triangle count (0 for inner)
};

struct Result {
    std::vector<WideNode> nodes; // the collapsed 4-wide tree (host form; the kernels read `quads`)
    std::vector<Pair> pairs;     // 2-wide full-precision records (same binary tree)
    std::vector<Pair> quads;     // 4-wide full-precision nodes of the collapsed tree: node j = records 2j (children 0, 1)
                                 // and 2j + 1 (children 2, 3); inner links are RECORD indices (2 x node), see build()
    int pair_depth = 0;          // depth of the pair tree (= binary depth - 1, root pair = 1)
    std::vector<int32_t> order;  // leaf order -> original triangle index
    int max_depth = 0;           // depth of the 4-wide tree (root = 1)
    int bin_depth = 0;           // depth of the binary tree it was collapsed from
    int num_leaves = 0;
    int stack_bound = 1;         // upper bound of the traversal stack: 3 entries per level
    bool ok = true;              // false if a leaf could not be referenced (more than 7 triangles)
};

// largest leaf the SAH may keep (a leaf reference carries the count in 3 bits); tunable for experiments: RT_BVH_MAX_LEAF
inline int max_leaf() {
    static int c = [] { const char *e = knob("RT_BVH_MAX_LEAF"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 7 ? 7 : v); }();
    return c;
}
#define kMaxLeaf (rtbvh::max_leaf())
constexpr int kTopPrefix = 1024;  // records laid out breadth-first at the front (LDS-cacheable top of the tree)
constexpr int kMaxBinDepth = 60;

// relative cost of a traversal step (tunable for experiments: RT_BVH_TRAV_COST)
inline float trav_cost() {
    static float c = [] { const char *e = knob("RT_BVH_TRAV_COST"); return e ? (float)atof(e) : 1.0f; }();
    return c;
}

inline float pad_down(float v, int k) {
    for (int i = 0; i < k; i++) v = std::nextafter(v, -FLT_MAX);
    return v;
}
inline float pad_up(float v, int k) {
    for (int i = 0; i < k; i++) v = std::nextafter(v, FLT_MAX);
    return v;
}

// ---- padding of the 4-wide records for the kernels' one-fma plane distance
// The 4-wide node step computes the distance to a plane b of axis a as ONE fused multiply-add, b * (1 / d_a) + s_a with
// s_a = -o_a * (1 / d_a) rounded on its own (half the vector instructions of (b - o_a) * (1 / d_a) for the 24 planes of a
// node).  The rounding of s_a moves EVERY plane of that axis, seen from the ray, by up to 2^-24 |o_a| in world units -- half an
// ulp of the ORIGIN's coordinate, whatever 1 / d_a is.  (The rounding of the fma itself and of 1 / d are relative to the
// distance and are covered, as before, by the widened exit distance of the test.)  So the records the kernels read are the
// 2-ulp-padded ones pushed outward by another 2^-23 R_a, where R_a bounds |o_a| over every ray that will be traced:
// origins of secondary rays lie on the scene's triangles (offset by at most 256 ulps: offset_ray_origin), the camera and
// the rays of the test hooks are looked at by the caller, who re-pads for a larger radius when one comes along.
inline void quads_abs_bounds(const std::vector<Pair> &quads, float m[3]) {
    m[0] = m[1] = m[2] = 0.f;
    for (const Pair &p : quads)
        for (int side = 0; side < 2; side++) {
            if ((side ? p.rlink : p.llink) == kNoChild) continue;
            const float *b = side ? p.rbox : p.lbox;
            for (int a = 0; a < 3; a++) {
                if (std::isfinite(b[a])) m[a] = std::max(m[a], std::fabs(b[a]));
                if (std::isfinite(b[3 + a])) m[a] = std::max(m[a], std::fabs(b[3 + a]));
            }
        }
    for (int a = 0; a < 3; a++) m[a] = m[a] * 1.001f;  // (256 ulps of offset and to spare)
}
inline void pad_quads_for_origins(const std::vector<Pair> &base, const float radius[3], std::vector<Pair> &out) {
    out = base;
    for (Pair &p : out)
        for (int side = 0; side < 2; side++) {
            if ((side ? p.rlink : p.llink) == kNoChild) continue;  // (all-+inf boxes: stay as they are)
            float *b = side ? p.rbox : p.lbox;
            for (int a = 0; a < 3; a++) {
                const double pad = std::ldexp((double)radius[a], -23);
                if (std::isfinite(b[a])) b[a] = pad_down((float)((double)b[a] - pad), 1);         // (the conversion rounds to nearest: one ulp further)
                if (std::isfinite(b[3 + a])) b[3 + a] = pad_up((float)((double)b[3 + a] + pad), 1);
            }
        }
}

// ---- step 1: binary SAH tree
inline void build_binary(const float *verts, int n, std::vector<BinNode> &bin, std::vector<int32_t> &order,
                         int &bin_depth, int &num_leaves) {
    bin.clear();
    order.assign(n, 0);
    bin_depth = 0;
    num_leaves = 0;
    if (n == 0) return;
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
        std::stable_sort(idx[a].begin(), idx[a].end(),
                         [&](int i, int j) { return cen[3 * (size_t)i + a] < cen[3 * (size_t)j + a]; });
    }
    std::vector<float> right_area(n);
    std::vector<uint8_t> side(n);
    std::vector<int32_t> tmp(n);
    struct Task {
        int node, begin, end, depth;
    };
    int out_pos = 0;
    Box root;
    root.reset();
    for (int i = 0; i < n; i++) root.extend(boxes[i]);
    bin.reserve(2 * (size_t)n);
    bin.push_back(BinNode());
    bin[0].box = root;
    std::vector<Task> stack;
    stack.push_back(Task{0, 0, n, 1});
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        const int cnt = t.end - t.begin;
        bin_depth = std::max(bin_depth, t.depth);
        int best_axis = -1, best_split = -1;
        if (cnt >= 2 && t.depth < kMaxBinDepth) {
            float best = FLT_MAX;
            for (int a = 0; a < 3; a++) {
                Box acc;
                acc.reset();
                for (int i = t.end - 1; i > t.begin; i--) {
                    acc.extend(boxes[idx[a][i]]);
                    right_area[i] = acc.half_area();
                }
                acc.reset();
                for (int i = t.begin; i < t.end - 1; i++) {
                    acc.extend(boxes[idx[a][i]]);
                    float cost = acc.half_area() * (float)(i + 1 - t.begin) + right_area[i + 1] * (float)(t.end - i - 1);
                    if (cost < best) {
                        best = cost;
                        best_axis = a;
                        best_split = i + 1;
                    }
                }
            }
            if (best_axis >= 0) {
                float area = bin[t.node].box.half_area();
                float leaf_cost = area * (float)cnt;
                float split_cost = area * trav_cost() + best;  // cost of one node step in triangle tests
                if (cnt <= kMaxLeaf && split_cost >= leaf_cost) best_axis = -1;
            }
            if (best_axis < 0 && cnt > kMaxLeaf) {  // coincident boxes: split in the middle of axis 0
                best_axis = 0;
                best_split = t.begin + cnt / 2;
            }
        }
        if (best_axis < 0) {  // leaf (a range longer than kMaxLeaf only at the depth cap)
            bin[t.node].first = out_pos;
            bin[t.node].count = cnt;
            for (int i = t.begin; i < t.end; i++) order[out_pos++] = idx[0][i];
            num_leaves++;
            continue;
        }
        for (int i = t.begin; i < best_split; i++) side[idx[best_axis][i]] = 0;
        for (int i = best_split; i < t.end; i++) side[idx[best_axis][i]] = 1;
        for (int a = 0; a < 3; a++) {
            if (a == best_axis) continue;
            int l = t.begin, r = 0;
            for (int i = t.begin; i < t.end; i++) {
                int tri = idx[a][i];
                if (side[tri] == 0) idx[a][l++] = tri;
                else tmp[r++] = tri;
            }
            memcpy(&idx[a][l], tmp.data(), sizeof(int32_t) * (size_t)r);
        }
        Box lb, rb;
        lb.reset();
        rb.reset();
        for (int i = t.begin; i < best_split; i++) lb.extend(boxes[idx[best_axis][i]]);
        for (int i = best_split; i < t.end; i++) rb.extend(boxes[idx[best_axis][i]]);
        int li = (int)bin.size();
        bin.push_back(BinNode());
        bin.push_back(BinNode());
        bin[li].box = lb;
        bin[li + 1].box = rb;
        bin[t.node].left = li;
        bin[t.node].right = li + 1;
        // depth-first, left first: a leaf's triangles and a subtree's leaves stay contiguous
        stack.push_back(Task{li + 1, best_split, t.end, t.depth + 1});
        stack.push_back(Task{li, t.begin, best_split, t.depth + 1});
    }
}

// ---- step 1b: insertion-based optimisation of the binary tree (after Bittner, Hapala, Havran 2013).
// The sweep builder decides top-down and never revisits a split; this pass takes subtrees out of the tree and puts each
// back where it adds the least surface area (branch-and-bound search over the whole tree), which lowers the SAH cost of
// the bunny scenes by 10 - 15 % and with it the node steps per ray.  Only the topology changes: leaves keep their
// triangle ranges, so the leaf order -- and, by the tie rule of the kernels, every traversal result -- stays the same.
// `passes`: sweeps over all nodes in order of decreasing surface area (RT_BVH_OPT, default 2; 0 = off).
inline int opt_passes() {
    static int c = [] { const char *e = knob("RT_BVH_OPT"); int v = e ? atoi(e) : 2; return v < 0 ? 0 : (v > 16 ? 16 : v); }();
    return c;
}
inline double sah_cost(const std::vector<BinNode> &bin) {
    double c = 0.0;
    for (const BinNode &b : bin) c += (double)b.box.half_area() * (b.left < 0 ? (double)b.count : (double)trav_cost());
    return bin.empty() ? 0.0 : c / std::max((double)bin[0].box.half_area(), 1e-30);
}
inline void optimize_reinsert(std::vector<BinNode> &bin, int passes) {
    const int n = (int)bin.size();
    if (n < 7 || passes <= 0) return;
    std::vector<int> parent(n, -1);
    for (int i = 0; i < n; i++)
        if (bin[i].left >= 0) {
            parent[bin[i].left] = i;
            parent[bin[i].right] = i;
        }
    auto unite = [](const Box &a, const Box &b) {
        Box r = a;
        r.extend(b);
        return r;
    };
    auto refit_up = [&](int i) {
        for (; i >= 0; i = parent[i]) {
            Box nb = unite(bin[bin[i].left].box, bin[bin[i].right].box);
            if (memcmp(&nb, &bin[i].box, sizeof(Box)) == 0) break;
            bin[i].box = nb;
        }
    };
    struct Cand {
        float induced;  // surface area added to the ancestors of `node` if the subtree is inserted below it
        int node;
        bool operator<(const Cand &o) const { return induced > o.induced; }  // min-heap on induced cost
    };
    std::vector<Cand> heap;
    std::vector<int> by_area(n);
    for (int pass = 0; pass < passes; pass++) {
        std::iota(by_area.begin(), by_area.end(), 0);
        std::stable_sort(by_area.begin(), by_area.end(),
                         [&](int a, int b) { return bin[a].box.half_area() > bin[b].box.half_area(); });
        for (int ci = 0; ci < n; ci++) {
            const int N = by_area[ci];
            const int P = parent[N];
            if (P <= 0) continue;  // the root and its children stay
            const int G = parent[P];
            const int S = bin[P].left == N ? bin[P].right : bin[P].left;
            // ---- take N (and its parent P) out: the sibling moves up
            if (bin[G].left == P) bin[G].left = S;
            else bin[G].right = S;
            parent[S] = G;
            refit_up(G);
            // ---- best place to put N back: branch and bound on the area it adds
            const Box nb = bin[N].box;
            const float na = nb.half_area();
            float best_cost = FLT_MAX;
            int best = S;
            heap.clear();
            heap.push_back(Cand{0.f, 0});
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const Cand c = heap.back();
                heap.pop_back();
                if (c.induced + na >= best_cost) break;  // nothing left in the heap can do better
                const BinNode &x = bin[c.node];
                const float direct = unite(x.box, nb).half_area();
                const float total = c.induced + direct;
                if (total < best_cost) {
                    best_cost = total;
                    best = c.node;
                }
                const float child_induced = total - x.box.half_area();
                if (x.left >= 0 && child_induced + na < best_cost) {
                    heap.push_back(Cand{child_induced, x.left});
                    std::push_heap(heap.begin(), heap.end());
                    heap.push_back(Cand{child_induced, x.right});
                    std::push_heap(heap.begin(), heap.end());
                }
            }
            // ---- P becomes the parent of {best, N} in best's old place
            const int X = best, XP = parent[X];
            if (XP < 0) {  // (above the root: keep index 0 the root -- swap contents instead)
                // the search never returns the root as the cheapest place unless the tree is degenerate; put N back beside S
                const int X2 = S, XP2 = parent[S];
                if (bin[XP2].left == X2) bin[XP2].left = P;
                else bin[XP2].right = P;
                parent[P] = XP2;
                bin[P].left = X2;
                bin[P].right = N;
                parent[X2] = P;
                parent[N] = P;
                refit_up(P);
                continue;
            }
            if (bin[XP].left == X) bin[XP].left = P;
            else bin[XP].right = P;
            parent[P] = XP;
            bin[P].left = X;
            bin[P].right = N;
            parent[X] = P;
            parent[N] = P;
            bin[P].box = unite(bin[X].box, nb);
            refit_up(XP);
        }
    }
}
// ---- step 2: collapse to 4-wide, write both record formats
// verts: n x 9 floats (p0 p1 p2).  Deterministic for a given input.
inline Result build(const float *verts, int n) {
    Result res;
    std::vector<BinNode> bin;
    build_binary(verts, n, bin, res.order, res.bin_depth, res.num_leaves);
    optimize_reinsert(bin, opt_passes());
    {  // depth of the (possibly re-shaped) binary tree
        std::vector<std::pair<int, int>> st;
        if (!bin.empty()) st.push_back({0, 1});
        res.bin_depth = bin.empty() ? 0 : 1;
        while (!st.empty()) {
            auto [i, d] = st.back();
            st.pop_back();
            res.bin_depth = std::max(res.bin_depth, d);
            if (bin[i].left >= 0) {
                st.push_back({bin[i].left, d + 1});
                st.push_back({bin[i].right, d + 1});
            }
        }
    }
    auto empty_node = [] {
        WideNode nd;
        for (int k = 0; k < 4; k++) {
            nd.link[k] = kNoChild;
            nd.box[k].reset();
        }
        return nd;
    };
    if (n == 0) {
        res.nodes.push_back(empty_node());
        Pair ep;
        memset(&ep, 0, sizeof(ep));
        ep.llink = ep.rlink = kNoChild;
        res.pairs.push_back(ep);
        res.quads.assign(2, absent_quad_record());
        res.max_depth = 1;
        res.pair_depth = 1;
        res.stack_bound = 1;
        return res;
    }
    struct Task {
        int bin_node, out_node, depth;
    };
    res.nodes.push_back(empty_node());
    std::vector<Task> stack;
    stack.push_back(Task{0, 0, 1});
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        res.max_depth = std::max(res.max_depth, t.depth);
        // children of the wide node: open the inner child with the largest surface area until 4
        int kids[4];
        int nk = 0;
        const BinNode &b = bin[t.bin_node];
        if (b.left < 0) {
            kids[nk++] = t.bin_node;  // the whole tree is one leaf
        } else {
            kids[nk++] = b.left;
            kids[nk++] = b.right;
            while (nk < 4) {
                int pick = -1;
                float best = -1.f;
                for (int k = 0; k < nk; k++)
                    if (bin[kids[k]].left >= 0 && bin[kids[k]].box.half_area() > best) {
                        best = bin[kids[k]].box.half_area();
                        pick = k;
                    }
                if (pick < 0) break;
                int opened = kids[pick];
                kids[pick] = bin[opened].left;
                kids[nk++] = bin[opened].right;
            }
        }
        WideNode nd = empty_node();
        for (int k = 0; k < nk; k++) nd.box[k] = bin[kids[k]].box;
        for (int k = 0; k < nk; k++) {
            const BinNode &c = bin[kids[k]];
            if (c.left < 0) {
                // a leaf longer than 7 triangles cannot be referenced (3 count bits); the builder only
                // produces such ranges at the binary depth cap, where they are split into several refs
                if (c.count > 7) res.ok = false;
                nd.link[k] = leaf_ref(c.first, std::min(c.count, 7));
            } else {
                int child = (int)res.nodes.size();
                res.nodes.push_back(empty_node());
                nd.link[k] = child;
                stack.push_back(Task{kids[k], child, t.depth + 1});
            }
        }
        res.nodes[t.out_node] = nd;
    }
    res.stack_bound = 3 * res.max_depth + 1;
    // ---- 2-wide records from the same binary tree
    {
        auto link_of = [&](int bn, std::vector<std::pair<int, int>> &todo, int &next) -> int32_t {
            const BinNode &c = bin[bn];
            if (c.left < 0) return leaf_ref(c.first, std::min(c.count, 7));
            int idx = next++;
            todo.push_back({bn, idx});
            return idx;
        };
        auto set_box = [&](float *dst, const Box &b) {
            for (int a = 0; a < 3; a++) {
                dst[a] = pad_down(b.lo[a], 2);
                dst[3 + a] = pad_up(b.hi[a], 2);
            }
        };
        std::vector<std::pair<int, int>> todo;  // (binary node, pair index), depth-first
        int next = 1;
        Pair empty;
        memset(&empty, 0, sizeof(empty));
        empty.llink = empty.rlink = kNoChild;
        res.pairs.assign(1, empty);
        if (bin[0].left < 0) {  // single leaf: left child = the leaf, right child empty
            set_box(res.pairs[0].lbox, bin[0].box);
            res.pairs[0].llink = leaf_ref(bin[0].first, std::min(bin[0].count, 7));
            res.pairs[0].rbox[0] = res.pairs[0].rbox[1] = res.pairs[0].rbox[2] = FLT_MAX;
            res.pairs[0].rbox[3] = res.pairs[0].rbox[4] = res.pairs[0].rbox[5] = -FLT_MAX;
            res.pair_depth = 1;
        } else {
            todo.push_back({0, 0});
            std::vector<int> depth_of(1, 1);
            while (!todo.empty()) {
                auto [bn, pi] = todo.back();
                todo.pop_back();
                if ((int)res.pairs.size() < next) res.pairs.resize(next, empty);
                if ((int)depth_of.size() < next) depth_of.resize(next, 0);
                int d = depth_of[pi];
                res.pair_depth = std::max(res.pair_depth, d);
                const BinNode &b = bin[bn];
                Pair p = empty;
                set_box(p.lbox, bin[b.left].box);
                set_box(p.rbox, bin[b.right].box);
                int before = next;
                p.rlink = link_of(b.right, todo, next);
                p.llink = link_of(b.left, todo, next);  // left pushed last -> processed first
                if ((int)res.pairs.size() < next) res.pairs.resize(next, empty);
                if ((int)depth_of.size() < next) depth_of.resize(next, 0);
                for (int k = before; k < next; k++) depth_of[k] = d + 1;
                res.pairs[pi] = p;
            }
        }
    }
    // ---- breadth-first prefix: the first kTopPrefix records of each format are the TOP of the tree in
    // level order (ancestor-closed), so a kernel can keep "the records with index < T" in LDS for any
    // T <= kTopPrefix; the remaining records keep their depth-first order.  Children still always have a
    // larger index than their parent.
    {
        {
            auto lk = [](const Pair &p, int *out) { out[0] = p.llink; out[1] = p.rlink; return 2; };
            const int n_rec = (int)res.pairs.size();
            if (n_rec > 2) {
                std::vector<int> new_of(n_rec, -1), order_new;
                std::vector<int> queue{0};
                new_of[0] = 0;
                order_new.push_back(0);
                for (size_t head = 0; head < queue.size() && (int)order_new.size() < kTopPrefix; head++) {
                    int links[2];
                    lk(res.pairs[queue[head]], links);
                    for (int k = 0; k < 2 && (int)order_new.size() < kTopPrefix; k++)
                        if (links[k] >= 0 && new_of[links[k]] < 0) {
                            new_of[links[k]] = (int)order_new.size();
                            order_new.push_back(links[k]);
                            queue.push_back(links[k]);
                        }
                }
                for (int i = 0; i < n_rec; i++)
                    if (new_of[i] < 0) {
                        new_of[i] = (int)order_new.size();
                        order_new.push_back(i);
                    }
                std::vector<Pair> old = res.pairs;
                for (int i = 0; i < n_rec; i++) {
                    Pair p = old[i];
                    if (p.llink >= 0) p.llink = new_of[p.llink];
                    if (p.rlink >= 0) p.rlink = new_of[p.rlink];
                    res.pairs[new_of[i]] = p;
                }
            }
        }
        {
            const int n_rec = (int)res.nodes.size();
            if (n_rec > 2) {
                std::vector<int> new_of(n_rec, -1), order_new;
                std::vector<int> queue{0};
                new_of[0] = 0;
                order_new.push_back(0);
                for (size_t head = 0; head < queue.size() && (int)order_new.size() < kTopPrefix; head++)
                    for (int k = 0; k < 4 && (int)order_new.size() < kTopPrefix; k++) {
                        int l = res.nodes[queue[head]].link[k];
                        if (l >= 0 && new_of[l] < 0) {
                            new_of[l] = (int)order_new.size();
                            order_new.push_back(l);
                            queue.push_back(l);
                        }
                    }
                for (int i = 0; i < n_rec; i++)
                    if (new_of[i] < 0) {
                        new_of[i] = (int)order_new.size();
                        order_new.push_back(i);
                    }
                std::vector<WideNode> old = res.nodes;
                for (int i = 0; i < n_rec; i++) {
                    WideNode nd = old[i];
                    for (int k = 0; k < 4; k++)
                        if (nd.link[k] >= 0) nd.link[k] = new_of[nd.link[k]];
                    res.nodes[new_of[i]] = nd;
                }
            }
        }
    }
    // ---- the same 4-wide tree with full-precision boxes: two pair-style records per node
    {
        const Pair empty = absent_quad_record();
        res.quads.assign(2 * res.nodes.size(), empty);
        for (size_t j = 0; j < res.nodes.size(); j++)
            for (int k = 0; k < 4; k++) {
                const int32_t link = res.nodes[j].link[k];
                if (link == kNoChild) continue;
                Pair &rec = res.quads[2 * j + (k >> 1)];
                float *dst = (k & 1) ? rec.rbox : rec.lbox;
                for (int a = 0; a < 3; a++) {
                    dst[a] = pad_down(res.nodes[j].box[k].lo[a], 2);
                    dst[3 + a] = pad_up(res.nodes[j].box[k].hi[a], 2);
                }
                ((k & 1) ? rec.rlink : rec.llink) = link >= 0 ? 2 * link : link;
            }
    }
    return res;
}

// The 4-wide format from ANY tree of pair records (the device LBVH builder emits `pairs` only): starting at the root,
// the inner child with the largest surface area is opened until a node has four children; nodes are numbered breadth-
// first.  Boxes are copied as they are (already padded).  Fills res.quads / max_depth / stack_bound.
inline void quads_from_pairs(Result &res) {
    const std::vector<Pair> &pairs = res.pairs;
    struct Child {
        int32_t link;
        const float *box;
    };
    auto area = [](const float *b) {
        const float e0 = b[3] - b[0], e1 = b[4] - b[1], e2 = b[5] - b[2];
        return (e0 + e1) * e2 + e0 * e1;
    };
    res.quads.clear();
    std::vector<int> pair_of{0}, depth_of{1};  // wide node -> the pair it was made from, its depth
    res.max_depth = 1;
    for (size_t j = 0; j < pair_of.size(); j++) {
        std::vector<Child> ch;
        auto add = [&](const Pair &p) {
            if (p.llink != kNoChild) ch.push_back({p.llink, p.lbox});
            if (p.rlink != kNoChild) ch.push_back({p.rlink, p.rbox});
        };
        add(pairs[(size_t)pair_of[j]]);
        while (ch.size() < 4) {
            int best = -1;
            for (size_t k = 0; k < ch.size(); k++)
                if (ch[k].link >= 0) {
                    const Pair &cp = pairs[(size_t)ch[k].link];
                    const int grows = (cp.llink != kNoChild) + (cp.rlink != kNoChild) - 1;
                    if ((int)ch.size() + grows <= 4 && (best < 0 || area(ch[k].box) > area(ch[(size_t)best].box))) best = (int)k;
                }
            if (best < 0) break;
            const Pair &cp = pairs[(size_t)ch[(size_t)best].link];
            ch.erase(ch.begin() + best);
            add(cp);
        }
        Pair rec[2] = {absent_quad_record(), absent_quad_record()};
        for (size_t k = 0; k < ch.size(); k++) {
            Pair &r = rec[k >> 1];
            memcpy((k & 1) ? r.rbox : r.lbox, ch[k].box, 6 * sizeof(float));
            int32_t link = ch[k].link;
            if (link >= 0) {  // becomes a wide node of its own
                pair_of.push_back(link);
                depth_of.push_back(depth_of[j] + 1);
                res.max_depth = std::max(res.max_depth, depth_of[j] + 1);
                link = 2 * (int32_t)(pair_of.size() - 1);
            }
            ((k & 1) ? r.rlink : r.llink) = link;
        }
        res.quads.push_back(rec[0]);
        res.quads.push_back(rec[1]);
    }
    res.stack_bound = 3 * res.max_depth + 1;
}

}  // namespace rtbvh
#endif  // RT_BVH_H
