// crt_bvh.h -- host BVH2 builder (binned SAH) for the closest-hit query of
// ComputeShader.wgsl:503-518.  The reference has no acceleration structure; the
// contract here is "return exactly what the loop over all primitives returns".
#pragma once
#include <cstdint>
#include <vector>

namespace crt {

// Child reference: >= 0 inner node index; < 0 leaf, ~ref = (first << 3) | (count - 1)
// where first indexes the leaf-ordered primitive array and 1 <= count <= 8.
#ifndef CRT_BVH_MAXLEAF
#define CRT_BVH_MAXLEAF 4
#endif
constexpr int kMaxLeaf = CRT_BVH_MAXLEAF;   // <= 8 (leaf encoding)
constexpr int kMaxDepth = 30;      // deepest leaf; the traversal stack holds 32
constexpr int kNodeFloats = 16;    // c0.lo c0.hi c1.lo c1.hi ref0 ref1 pad pad  (64 B)

struct Bvh {
    std::vector<float> nodes;      // kNodeFloats per inner node
    std::vector<uint32_t> order;   // leaf slot -> original primitive position
    int32_t root = -1;             // child reference of the root
    uint32_t n_inner = 0, n_leaves = 0, max_depth = 0;
};

// lo/hi: n x 3 floats, already padded conservatively by the caller.
void build_bvh2(const float *lo, const float *hi, uint32_t n, Bvh &out);

// 4-wide form of the same tree for the wavefront traversal kernel: one 128-byte line per
// node (random gathers cost per REQUEST on gfx950, not per byte -- see DESIGN.md 5.1):
//   floats [0..3] lo.x of children 0..3, [4..7] lo.y, [8..11] lo.z, [12..15] hi.x, [16..19] hi.y,
//   [20..23] hi.z, [24..27] child refs (same encoding as Bvh), [28..31] unused.
// Empty child slots hold a far-away point box (lo = hi = 3e38, missed by every ray) and ref 0.
constexpr int kNode4Floats = 32;
struct Bvh4 {
    std::vector<float> nodes;
    int32_t root = -1;
    uint32_t n_inner = 0, max_depth = 0;
};
void collapse_bvh4(const Bvh &b2, Bvh4 &out);

// Quantised form of Bvh4: 64 bytes per node.  Box planes are 16-bit grid coordinates over the
// scene bounds, rounded outward with one extra grid unit of slack (culling stays conservative):
//   u16 lo.x[4] lo.y[4] | lo.z[4] hi.x[4] | hi.y[4] hi.z[4] | i32 refs[4]      (4 x 16 B)
// plane = base[axis] + q * scale[axis].  Empty slots: lo = 65535, hi = 0 (never hit).
// `ok` is false when the scene cannot be quantised safely (unbounded primitives, or bounds
// so far from the origin that f32 rounding would exceed the slack); use Bvh4 then.
struct Bvh4Q {
    std::vector<uint32_t> nodes;   // 16 dwords per node
    float base[3] = {0, 0, 0}, scale[3] = {1, 1, 1};
    bool ok = false;
};
void quantize_bvh4(const Bvh4 &b4, Bvh4Q &out);

// 8-wide quantised form, collapsed straight from the BVH2: 128 bytes per node = one L2 line, so a node
// costs one memory request like a 64-byte one but a ray visits ~1/3 fewer of them (the traversal kernel
// runs at the memory system's random-request rate, not at its bandwidth -- DESIGN.md 5.1).
//   dwords  0.. 3  lo.x of children 0..7 (two 16-bit grid coordinates per dword, child 2k in the low half)
//           4.. 7  lo.y     8..11  lo.z    12..15  hi.x    16..19  hi.y    20..23  hi.z
//          24..31  child references (>= 0: 8-wide node index, < 0: ~(first_slot << 3 | count-1) as in the BVH2)
// Same grid, slack and empty-slot convention as Bvh4Q; nodes are numbered breadth-first.
struct Bvh8Q {
    std::vector<uint32_t> nodes;   // 32 dwords per node
    float base[3] = {0, 0, 0}, scale[3] = {1, 1, 1};
    int32_t root = -1;
    uint32_t n_inner = 0, max_depth = 0;
    bool ok = false;
};
void build_bvh8q(const Bvh &b2, Bvh8Q &out);

// Result of the all-device LBVH build (crt_lbvh.hip build_lbvh_device): the arrays themselves stay on the device.
struct LbvhDeviceResult {
    uint32_t n_nodes4 = 0;         // 4-wide nodes written (breadth-first numbering, root = 0)
    uint32_t max_depth = 0;        // of the BVH2
    float qbase[3] = {0, 0, 0}, qscale[3] = {1, 1, 1};
    bool quantised = false;        // false: the scene cannot be quantised (quantize_bvh4's rules); use the host path
};

}  // namespace crt
