// crt_bvh.h -- host BVH2 builder (binned SAH) for the closest-hit query of
// ComputeShader.wgsl:503-518.  The reference has no acceleration structure; the
// contract here is "return exactly what the loop over all primitives returns".
#pragma once
#include <cstdint>
#include <vector>

namespace crt {

// Child reference: >= 0 inner node index; < 0 leaf, ~ref = (first << 3) | (count - 1)
// where first indexes the leaf-ordered primitive array and 1 <= count <= 8.
constexpr int kMaxLeaf = 4;
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

}  // namespace crt
