// crt_bvh.cpp -- binned-SAH BVH2 build on the host (multi-threaded over subtrees).
#include "crt_bvh.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <thread>

namespace crt {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { lo[0] = lo[1] = lo[2] = FLT_MAX; hi[0] = hi[1] = hi[2] = -FLT_MAX; }
    void grow(const float *l, const float *h) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    void grow(const Box &b) { grow(b.lo, b.hi); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

#ifndef CRT_BVH_BINS
#define CRT_BVH_BINS 32
#endif
#ifndef CRT_BVH_COST_TRAV
#define CRT_BVH_COST_TRAV 1.2f
#endif
constexpr int kBins = CRT_BVH_BINS;
constexpr float kCostTrav = CRT_BVH_COST_TRAV;   // one inner visit (two boxes, 64 B) vs one primitive test
constexpr float kCostIsect = 1.0f;

struct Builder {
    const float *lo, *hi;
    uint32_t n;
    std::vector<uint32_t> &order;
    std::vector<float> cent;            // n x 3 centroids (of the boxes)
    std::vector<float> &nodes;
    std::atomic<uint32_t> next_node{0};
    std::atomic<uint32_t> n_leaves{0};
    std::atomic<uint32_t> max_depth{0};
    std::atomic<int> threads_left;

    Builder(const float *l, const float *h, uint32_t n_, std::vector<uint32_t> &ord, std::vector<float> &nd)
        : lo(l), hi(h), n(n_), order(ord), nodes(nd), threads_left(0) {}

    int32_t make_leaf(uint32_t begin, uint32_t count, uint32_t depth) {
        n_leaves++;
        uint32_t d = max_depth.load();
        while (depth > d && !max_depth.compare_exchange_weak(d, depth)) {}
        return ~(int32_t)((begin << 3) | (count - 1));
    }

    // Builds [begin,end) of `order`; returns the child reference and its bounds.
    int32_t build(uint32_t begin, uint32_t end, uint32_t depth, Box &bounds) {
        uint32_t count = end - begin;
        bounds.reset();
        Box cb; cb.reset();
        for (uint32_t i = begin; i < end; i++) {
            uint32_t p = order[i];
            bounds.grow(lo + 3 * p, hi + 3 * p);
            const float *c = &cent[3 * p];
            cb.grow(c, c);
        }
        if (count == 1) return make_leaf(begin, 1, depth);

        // Capacity that keeps every leaf at depth <= kMaxDepth.
        int levels_left = kMaxDepth - (int)depth;          // levels available below this node
        uint64_t child_cap = (levels_left >= 1) ? ((uint64_t)kMaxLeaf << std::min(levels_left - 1, 40)) : 0;

        int best_axis = -1, best_bin = -1;
        float best_cost = FLT_MAX;
        float ext[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
        if (levels_left >= 1) {
            for (int axis = 0; axis < 3; axis++) {
                if (!(ext[axis] > 0.0f)) continue;
                Box bb[kBins]; uint32_t bc[kBins];
                for (int b = 0; b < kBins; b++) { bb[b].reset(); bc[b] = 0; }
                float scale = (float)kBins / ext[axis];
                for (uint32_t i = begin; i < end; i++) {
                    uint32_t p = order[i];
                    int b = (int)((cent[3 * p + axis] - cb.lo[axis]) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bb[b].grow(lo + 3 * p, hi + 3 * p);
                    bc[b]++;
                }
                float right_area[kBins]; uint32_t right_cnt[kBins];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    if (bc[b]) acc.grow(bb[b]);
                    c += bc[b];
                    right_area[b] = c ? acc.half_area() : 0.0f;
                    right_cnt[b] = c;
                }
                acc.reset(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    if (bc[b]) acc.grow(bb[b]);
                    c += bc[b];
                    uint32_t rc = right_cnt[b + 1];
                    if (c == 0 || rc == 0) continue;
                    if (c > child_cap || rc > child_cap) continue;
                    float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)rc;
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        }
        float parent_area = bounds.half_area();
        if (count <= (uint32_t)kMaxLeaf) {
            float leaf_cost = kCostIsect * (float)count;
            float split_cost = (best_axis >= 0 && parent_area > 0.0f)
                                   ? kCostTrav + kCostIsect * best_cost / parent_area : FLT_MAX;
            if (!(split_cost < leaf_cost) || levels_left < 1) return make_leaf(begin, count, depth);
        }
        uint32_t mid;
        if (best_axis >= 0) {
            float scale = (float)kBins / ext[best_axis];
            float clo = cb.lo[best_axis];
            int axis = best_axis, bin = best_bin;
            auto it = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t p) {
                int b = (int)((cent[3 * p + axis] - clo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= bin;
            });
            mid = (uint32_t)(it - order.begin());
        } else {
            // degenerate centroids or no admissible SAH split: object median on the widest axis
            int axis = 0;
            if (ext[1] > ext[axis]) axis = 1;
            if (ext[2] > ext[axis]) axis = 2;
            mid = begin + count / 2;
            std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end,
                             [&](uint32_t a, uint32_t b) {
                                 float ca = cent[3 * a + axis], cb2 = cent[3 * b + axis];
                                 return ca < cb2 || (ca == cb2 && a < b);
                             });
        }
        if (mid == begin || mid == end) mid = begin + count / 2;

        uint32_t node = next_node.fetch_add(1);
        Box b0, b1;
        int32_t r0, r1;
        bool spawn = count > 200000 && threads_left.fetch_sub(1) > 0;
        if (spawn) {
            std::thread t([&] { r0 = build(begin, mid, depth + 1, b0); });
            r1 = build(mid, end, depth + 1, b1);
            t.join();
            threads_left.fetch_add(1);
        } else {
            if (count > 200000) threads_left.fetch_add(1);
            r0 = build(begin, mid, depth + 1, b0);
            r1 = build(mid, end, depth + 1, b1);
        }
        float *nd = &nodes[(size_t)node * kNodeFloats];
        for (int a = 0; a < 3; a++) {
            nd[a] = b0.lo[a]; nd[3 + a] = b0.hi[a];
            nd[6 + a] = b1.lo[a]; nd[9 + a] = b1.hi[a];
        }
        std::memcpy(&nd[12], &r0, 4);
        std::memcpy(&nd[13], &r1, 4);
        nd[14] = 0.0f; nd[15] = 0.0f;
        return (int32_t)node;
    }
};

}  // namespace

void build_bvh2(const float *lo, const float *hi, uint32_t n, Bvh &out)
{
    out = Bvh();
    out.order.resize(n);
    for (uint32_t i = 0; i < n; i++) out.order[i] = i;
    if (n == 0) { out.root = -1; return; }
    out.nodes.assign((size_t)std::max<uint32_t>(n, 1) * kNodeFloats, 0.0f);
    Builder b(lo, hi, n, out.order, out.nodes);
    b.cent.resize((size_t)n * 3);
    for (size_t i = 0; i < (size_t)n * 3; i++) b.cent[i] = 0.5f * lo[i] + 0.5f * hi[i];
    unsigned hw = std::thread::hardware_concurrency();
    b.threads_left = (int)std::min<unsigned>(hw ? hw : 1, 32) - 1;
    Box root_bounds;
    out.root = b.build(0, n, 0, root_bounds);
    out.n_inner = b.next_node.load();
    out.n_leaves = b.n_leaves.load();
    out.max_depth = b.max_depth.load();
    out.nodes.resize((size_t)std::max<uint32_t>(out.n_inner, 1) * kNodeFloats);
}


namespace {
struct Child4 { float lo[3], hi[3]; int32_t ref; };
float area_of(const Child4 &c)
{
    float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
void children_of(const Bvh &b2, int32_t node, Child4 out[2])
{
    const float *nd = &b2.nodes[(size_t)node * kNodeFloats];
    for (int c = 0; c < 2; c++) {
        for (int a = 0; a < 3; a++) { out[c].lo[a] = nd[6 * c + a]; out[c].hi[a] = nd[6 * c + 3 + a]; }
        std::memcpy(&out[c].ref, &nd[12 + c], 4);
    }
}
int32_t collapse_rec(const Bvh &b2, int32_t ref2, Bvh4 &out, uint32_t depth)
{
    if (ref2 < 0) { out.max_depth = std::max(out.max_depth, depth); return ref2; }
    Child4 ch[4];
    int n = 2;
    children_of(b2, ref2, ch);
    while (n < 4) {                       // open the inner child with the largest surface
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < n; i++) if (ch[i].ref >= 0) { float a = area_of(ch[i]); if (a > ba) { ba = a; best = i; } }
        if (best < 0) break;
        Child4 two[2];
        children_of(b2, ch[best].ref, two);
        ch[best] = two[0];
        ch[n++] = two[1];
    }
    const uint32_t node = out.n_inner++;
    if (out.nodes.size() < (size_t)(node + 1) * kNode4Floats) out.nodes.resize(std::max<size_t>(out.nodes.size() * 2, (size_t)(node + 1) * kNode4Floats));
    int32_t refs[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) refs[i] = collapse_rec(b2, ch[i].ref, out, depth + 1);
    float *nd = &out.nodes[(size_t)node * kNode4Floats];
    for (int i = 0; i < 4; i++) {
        for (int a = 0; a < 3; a++) {
            nd[4 * a + i] = i < n ? ch[i].lo[a] : 3.0e38f;      // empty slot: a far-away point box,
            nd[12 + 4 * a + i] = i < n ? ch[i].hi[a] : 3.0e38f;  // missed by every ray in the slab test
        }
        std::memcpy(&nd[24 + i], &refs[i], 4);
        nd[28 + i] = 0.0f;
    }
    return (int32_t)node;
}
}  // namespace

void collapse_bvh4(const Bvh &b2, Bvh4 &out)
{
    out = Bvh4();
    out.nodes.assign((size_t)std::max<uint32_t>(b2.n_inner / 2 + 1, 1) * kNode4Floats, 0.0f);
    out.root = collapse_rec(b2, b2.root, out, 0);
    out.nodes.resize((size_t)std::max<uint32_t>(out.n_inner, 1) * kNode4Floats);
    // Renumber breadth-first: the top levels become nodes [0, N) (cached in LDS by the traversal
    // kernel) and siblings' subtrees sit near each other.
    if (out.root >= 0 && out.n_inner > 1) {
        std::vector<int32_t> order; order.reserve(out.n_inner);      // order[new] = old
        std::vector<int32_t> newid(out.n_inner, -1);
        order.push_back(out.root); newid[out.root] = 0;
        for (size_t head = 0; head < order.size(); head++) {
            const float *nd = &out.nodes[(size_t)order[head] * kNode4Floats];
            for (int i = 0; i < 4; i++) {
                int32_t r; std::memcpy(&r, &nd[24 + i], 4);
                if (nd[i] < 3.0e38f && r >= 0 && newid[r] < 0) { newid[r] = (int32_t)order.size(); order.push_back(r); }
            }
        }
        std::vector<float> nn(out.nodes.size());
        for (size_t k = 0; k < order.size(); k++) {
            const float *src = &out.nodes[(size_t)order[k] * kNode4Floats];
            float *dst = &nn[k * kNode4Floats];
            std::memcpy(dst, src, kNode4Floats * sizeof(float));
            for (int i = 0; i < 4; i++) {
                int32_t r; std::memcpy(&r, &src[24 + i], 4);
                if (src[i] < 3.0e38f && r >= 0) { r = newid[r]; std::memcpy(&dst[24 + i], &r, 4); }
            }
        }
        out.nodes.swap(nn);
        out.root = 0;
    }
}


void quantize_bvh4(const Bvh4 &b4, Bvh4Q &out)
{
    out = Bvh4Q();
    if (b4.n_inner == 0) return;
    float glo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, ghi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t n = 0; n < b4.n_inner; n++) {
        const float *nd = &b4.nodes[(size_t)n * kNode4Floats];
        for (int i = 0; i < 4; i++) {
            if (nd[i] >= 3.0e38f) continue;                         // empty slot
            for (int a = 0; a < 3; a++) {
                float l = nd[4 * a + i], h = nd[12 + 4 * a + i];
                if (!(l > -1.0e30f) || !(h < 1.0e30f)) return;      // unbounded primitive: not quantisable
                glo[a] = std::min(glo[a], l); ghi[a] = std::max(ghi[a], h);
            }
        }
    }
    for (int a = 0; a < 3; a++) {
        float ext = std::max(ghi[a] - glo[a], 1.0e-3f);
        float mag = std::max(std::fabs(glo[a]), std::fabs(ghi[a]));
        if (mag > 16.0f * ext) return;                              // too far from the origin for the slack
        out.base[a] = glo[a];
        out.scale[a] = ext / 65533.0f;
    }
    out.nodes.resize((size_t)b4.n_inner * 16);
    for (uint32_t n = 0; n < b4.n_inner; n++) {
        const float *nd = &b4.nodes[(size_t)n * kNode4Floats];
        uint16_t q[24];
        for (int i = 0; i < 4; i++) {
            const bool empty = nd[i] >= 3.0e38f;
            for (int a = 0; a < 3; a++) {
                if (empty) { q[4 * a + i] = 65535; q[12 + 4 * a + i] = 0; continue; }
                double l = ((double)nd[4 * a + i] - out.base[a]) / out.scale[a];
                double h = ((double)nd[12 + 4 * a + i] - out.base[a]) / out.scale[a];
                long ql = (long)std::floor(l) - 1, qh = (long)std::ceil(h) + 1;
                q[4 * a + i] = (uint16_t)std::min<long>(std::max<long>(ql, 0), 65535);
                q[12 + 4 * a + i] = (uint16_t)std::min<long>(std::max<long>(qh, 0), 65535);
            }
        }
        uint32_t *o = &out.nodes[(size_t)n * 16];
        for (int k = 0; k < 12; k++) o[k] = (uint32_t)q[2 * k] | ((uint32_t)q[2 * k + 1] << 16);
        std::memcpy(&o[12], &nd[24], 16);
    }
    out.ok = true;
}


namespace {
struct Node8 { Child4 ch[8]; int n; };
int32_t collapse8_rec(const Bvh &b2, int32_t ref2, std::vector<Node8> &nodes, uint32_t depth, uint32_t &max_depth)
{
    if (ref2 < 0) { max_depth = std::max(max_depth, depth); return ref2; }
    Node8 nd;
    nd.n = 2;
    children_of(b2, ref2, nd.ch);
    while (nd.n < 8) {                    // open the inner child with the largest surface
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < nd.n; i++) if (nd.ch[i].ref >= 0) { float a = area_of(nd.ch[i]); if (a > ba) { ba = a; best = i; } }
        if (best < 0) break;
        Child4 two[2];
        children_of(b2, nd.ch[best].ref, two);
        nd.ch[best] = two[0];
        nd.ch[nd.n++] = two[1];
    }
    const size_t me = nodes.size();
    nodes.push_back(nd);
    for (int i = 0; i < nd.n; i++) {
        const int32_t r = collapse8_rec(b2, nd.ch[i].ref, nodes, depth + 1, max_depth);
        nodes[me].ch[i].ref = r;
    }
    return (int32_t)me;
}
}  // namespace

void build_bvh8q(const Bvh &b2, Bvh8Q &out)
{
    out = Bvh8Q();
    if (b2.root < 0 || b2.n_inner == 0) return;
    std::vector<Node8> nodes;
    nodes.reserve(b2.n_inner / 3 + 16);
    uint32_t max_depth = 0;
    const int32_t root = collapse8_rec(b2, b2.root, nodes, 0, max_depth);
    if (root < 0) return;
    // scene bounds / quantisability (same rules as quantize_bvh4)
    float glo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, ghi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (const Node8 &nd : nodes)
        for (int i = 0; i < nd.n; i++)
            for (int a = 0; a < 3; a++) {
                const float l = nd.ch[i].lo[a], h = nd.ch[i].hi[a];
                if (!(l > -1.0e30f) || !(h < 1.0e30f)) return;      // unbounded primitive: not quantisable
                glo[a] = std::min(glo[a], l); ghi[a] = std::max(ghi[a], h);
            }
    for (int a = 0; a < 3; a++) {
        const float ext = std::max(ghi[a] - glo[a], 1.0e-3f);
        const float mag = std::max(std::fabs(glo[a]), std::fabs(ghi[a]));
        if (mag > 16.0f * ext) return;                              // too far from the origin for the slack
        out.base[a] = glo[a];
        out.scale[a] = ext / 65533.0f;
    }
    // breadth-first numbering
    std::vector<int32_t> order; order.reserve(nodes.size());        // order[new] = old
    std::vector<int32_t> newid(nodes.size(), -1);
    order.push_back(root); newid[root] = 0;
    for (size_t head = 0; head < order.size(); head++) {
        const Node8 &nd = nodes[order[head]];
        for (int i = 0; i < nd.n; i++) {
            const int32_t r = nd.ch[i].ref;
            if (r >= 0 && newid[r] < 0) { newid[r] = (int32_t)order.size(); order.push_back(r); }
        }
    }
    out.nodes.assign(order.size() * 32, 0u);
    for (size_t k = 0; k < order.size(); k++) {
        const Node8 &nd = nodes[order[k]];
        uint16_t q[48];                                             // [plane 0..5][child 0..7]
        int32_t refs[8];
        for (int i = 0; i < 8; i++) {
            const bool empty = i >= nd.n;
            refs[i] = empty ? 0 : (nd.ch[i].ref >= 0 ? newid[nd.ch[i].ref] : nd.ch[i].ref);
            for (int a = 0; a < 3; a++) {
                if (empty) { q[8 * a + i] = 65535; q[24 + 8 * a + i] = 0; continue; }
                const double l = ((double)nd.ch[i].lo[a] - out.base[a]) / out.scale[a];
                const double h = ((double)nd.ch[i].hi[a] - out.base[a]) / out.scale[a];
                const long ql = (long)std::floor(l) - 1, qh = (long)std::ceil(h) + 1;
                q[8 * a + i] = (uint16_t)std::min<long>(std::max<long>(ql, 0), 65535);
                q[24 + 8 * a + i] = (uint16_t)std::min<long>(std::max<long>(qh, 0), 65535);
            }
        }
        uint32_t *o = &out.nodes[k * 32];
        for (int d = 0; d < 24; d++) o[d] = (uint32_t)q[2 * d] | ((uint32_t)q[2 * d + 1] << 16);
        std::memcpy(&o[24], refs, 32);
    }
    out.root = 0;
    out.n_inner = (uint32_t)order.size();
    out.max_depth = max_depth;
    out.ok = true;
}

}  // namespace crt
