// crt_lbvh.hip -- BVH2 construction on the GPU: Morton codes, radix sort, Karras'
// parallel hierarchy, bottom-up bounds.  Produces the same `Bvh` structure as the host's binned-SAH
// builder (crt_bvh.cpp), so everything downstream -- collapse to 4-wide, quantisation, every
// kernel -- is shared, and so is the result: the accept rule of hit_test makes the closest hit
// independent of the tree, only the number of nodes a ray visits differs (an LBVH is
// a worse tree than a SAH one; what it buys is build time: milliseconds instead of seconds).
//
//   1. key_i   = morton30(centroid_i in the scene box) << 32 | i          (unique, so no ties)
//   2. sort keys (hipcub radix sort, 62 significant bits)
//   3. one thread per inner node i in [0, n-2]: the range of keys sharing its prefix and the split
//      position, from the longest-common-prefix function delta(i,j) = clz(key_i ^ key_j)
//      (T. Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012)
//   4. one thread per leaf climbs to the root; the second thread to arrive at a node (atomic flag)
//      unions the children's boxes and goes on
// Leaves hold one primitive each (reference ~(slot << 3 | 0), slot = sorted position).  Depth is at
// most 62 (one key bit per level), inside the 64-entry stacks of the single-ray walk.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>

#include "crt_bvh.h"

namespace crt {
namespace {

__device__ __forceinline__ uint32_t expand10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_lbvh_keys(const float *__restrict__ lo, const float *__restrict__ hi, uint32_t n,
                                                   float bx, float by, float bz, float sx, float sy, float sz,
                                                   unsigned long long *__restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float cx = 0.5f * lo[3 * (size_t)i + 0] + 0.5f * hi[3 * (size_t)i + 0];
    const float cy = 0.5f * lo[3 * (size_t)i + 1] + 0.5f * hi[3 * (size_t)i + 1];
    const float cz = 0.5f * lo[3 * (size_t)i + 2] + 0.5f * hi[3 * (size_t)i + 2];
    // (unbounded primitives have centroid +-inf or nan: clamped into the grid, any cell will do)
    const float fx = (cx - bx) * sx, fy = (cy - by) * sy, fz = (cz - bz) * sz;
    const uint32_t qx = fx > 0.0f ? (fx < 1023.0f ? (uint32_t)fx : 1023u) : 0u;
    const uint32_t qy = fy > 0.0f ? (fy < 1023.0f ? (uint32_t)fy : 1023u) : 0u;
    const uint32_t qz = fz > 0.0f ? (fz < 1023.0f ? (uint32_t)fz : 1023u) : 0u;
    const uint32_t m = (expand10(qx) << 2) | (expand10(qy) << 1) | expand10(qz);
    keys[i] = ((unsigned long long)m << 32) | i;
}

__device__ __forceinline__ int delta(const unsigned long long *__restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));        // keys are unique: never 64
}

// children of inner node i: references >= 0 are inner nodes, < 0 are leaves ~(slot << 3)
__global__ __launch_bounds__(256) void k_lbvh_hierarchy(const unsigned long long *__restrict__ keys, int n,
                                                        int *__restrict__ child, int *__restrict__ parent_inner, int *__restrict__ parent_leaf)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int g = i + s * d + (d < 0 ? d : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const int left = (first == g) ? ~(g << 3) : g;
    const int right = (last == g + 1) ? ~((g + 1) << 3) : g + 1;
    child[2 * i + 0] = left; child[2 * i + 1] = right;
    if (left >= 0) parent_inner[left] = i; else parent_leaf[g] = i;
    if (right >= 0) parent_inner[right] = i; else parent_leaf[g + 1] = i;
    if (i == 0) parent_inner[0] = -1;
}

__device__ __forceinline__ void child_box(int ref, const unsigned long long *__restrict__ keys, const float *__restrict__ lo,
                                          const float *__restrict__ hi, const float *__restrict__ nb, float b[6])
{
    if (ref < 0) {
        const uint32_t prim = (uint32_t)(keys[(~ref) >> 3] & 0xFFFFFFFFull);
        for (int a = 0; a < 3; a++) { b[a] = lo[3 * (size_t)prim + a]; b[3 + a] = hi[3 * (size_t)prim + a]; }
    } else {
        for (int a = 0; a < 6; a++) b[a] = __builtin_nontemporal_load(&nb[6 * (size_t)ref + a]);
    }
}

// One thread per leaf; the second arrival at a node writes its record (the children's boxes, the
// layout of crt_bvh.h) and its own box, then goes on to the parent.
__global__ __launch_bounds__(256) void k_lbvh_bounds(const unsigned long long *__restrict__ keys, int n, const float *__restrict__ lo,
                                                     const float *__restrict__ hi, const int *__restrict__ child,
                                                     const int *__restrict__ parent_inner, const int *__restrict__ parent_leaf,
                                                     unsigned int *__restrict__ flag, float *__restrict__ nb, float *__restrict__ nodes)
{
    const int leaf = (int)(blockIdx.x * 256u + threadIdx.x);
    if (leaf >= n) return;
    int cur = parent_leaf[leaf];
    while (cur >= 0) {
        __threadfence();                                    // my writes (a child's box) before the flag
        if (atomicAdd(&flag[cur], 1u) == 0u) return;       // first to arrive: the sibling's subtree is not ready
        __threadfence();
        float b0[6], b1[6];
        const int c0 = child[2 * cur], c1 = child[2 * cur + 1];
        child_box(c0, keys, lo, hi, nb, b0);
        child_box(c1, keys, lo, hi, nb, b1);
        float *nd = nodes + (size_t)cur * kNodeFloats;
        for (int a = 0; a < 6; a++) { nd[a] = b0[a]; nd[6 + a] = b1[a]; }
        nd[12] = __int_as_float(c0); nd[13] = __int_as_float(c1); nd[14] = 0.0f; nd[15] = 0.0f;
        for (int a = 0; a < 3; a++) {
            // (min/max that let a NaN bound through would poison every ancestor: an unbounded primitive has +-3e38 here)
            __builtin_nontemporal_store(fminf(b0[a], b1[a]), &nb[6 * (size_t)cur + a]);
            __builtin_nontemporal_store(fmaxf(b0[3 + a], b1[3 + a]), &nb[6 * (size_t)cur + 3 + a]);
        }
        cur = parent_inner[cur];
    }
}

template <typename T>
struct Tmp {
    T *p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)); }
    ~Tmp() { if (p) (void)hipFree(p); }
};

}  // namespace

// lo/hi: n x 3 floats on the host (padded conservatively by the caller, finite: unbounded
// primitives come in as +-3e38).  Needs n >= 2.
hipError_t build_lbvh(const float *lo, const float *hi, uint32_t n, Bvh &out, hipStream_t stream)
{
    out = Bvh();
    if (n < 2) return hipErrorInvalidValue;
    // scene box of the centroids (host: one pass over data that is in cache from the bounds computation)
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * lo[3 * i + a] + 0.5f * hi[3 * i + a];
            if (c > -1.0e30f && c < 1.0e30f) { clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c); }
        }
    float sc[3];
    for (int a = 0; a < 3; a++) {
        if (!(clo[a] <= chi[a])) { clo[a] = 0.0f; chi[a] = 1.0f; }
        sc[a] = 1024.0f / std::max(chi[a] - clo[a], 1.0e-20f);
    }
    Tmp<float> d_lo, d_hi, d_nb, d_nodes;
    Tmp<unsigned long long> d_keys, d_sorted;
    Tmp<int> d_child, d_pi, d_pl;
    Tmp<unsigned int> d_flag;
    Tmp<char> d_tmp;
    hipError_t e;
#define LB(call) do { e = (call); if (e != hipSuccess) return e; } while (0)
    LB(d_lo.alloc((size_t)n * 3)); LB(d_hi.alloc((size_t)n * 3));
    LB(d_keys.alloc(n)); LB(d_sorted.alloc(n));
    LB(d_child.alloc((size_t)2 * (n - 1))); LB(d_pi.alloc(n - 1)); LB(d_pl.alloc(n));
    LB(d_flag.alloc(n - 1)); LB(d_nb.alloc((size_t)6 * (n - 1))); LB(d_nodes.alloc((size_t)(n - 1) * kNodeFloats));
    LB(hipMemcpyAsync(d_lo.p, lo, (size_t)n * 12, hipMemcpyHostToDevice, stream));
    LB(hipMemcpyAsync(d_hi.p, hi, (size_t)n * 12, hipMemcpyHostToDevice, stream));
    LB(hipMemsetAsync(d_flag.p, 0, (size_t)(n - 1) * 4, stream));
    const unsigned blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_lbvh_keys, dim3(blocks), dim3(256), 0, stream, d_lo.p, d_hi.p, n, clo[0], clo[1], clo[2], sc[0], sc[1], sc[2], d_keys.p);
    LB(hipGetLastError());
    size_t tmp_bytes = 0;
    LB(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys.p, d_sorted.p, (int)n, 0, 62, stream));
    LB(d_tmp.alloc(tmp_bytes));
    LB(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tmp_bytes, d_keys.p, d_sorted.p, (int)n, 0, 62, stream));
    hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(blocks), dim3(256), 0, stream, d_sorted.p, (int)n, d_child.p, d_pi.p, d_pl.p);
    LB(hipGetLastError());
    hipLaunchKernelGGL(k_lbvh_bounds, dim3(blocks), dim3(256), 0, stream, d_sorted.p, (int)n, d_lo.p, d_hi.p, d_child.p, d_pi.p, d_pl.p,
                       d_flag.p, d_nb.p, d_nodes.p);
    LB(hipGetLastError());
    out.nodes.resize((size_t)(n - 1) * kNodeFloats);
    std::vector<unsigned long long> keys(n);
    LB(hipMemcpyAsync(out.nodes.data(), d_nodes.p, out.nodes.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    LB(hipMemcpyAsync(keys.data(), d_sorted.p, (size_t)n * 8, hipMemcpyDeviceToHost, stream));
    LB(hipStreamSynchronize(stream));
#undef LB
    out.order.resize(n);
    for (size_t s = 0; s < n; s++) out.order[s] = (uint32_t)(keys[s] & 0xFFFFFFFFull);
    out.root = 0;
    out.n_inner = n - 1;
    out.n_leaves = n;
    // depth (statistics; the walk's stacks hold 64 entries, the tree has at most 62 levels)
    {
        std::vector<std::pair<int32_t, uint32_t>> st;
        st.emplace_back(0, 1u);
        uint32_t md = 0;
        while (!st.empty()) {
            const auto [node, dep] = st.back(); st.pop_back();
            md = std::max(md, dep);
            for (int c = 0; c < 2; c++) {
                int32_t r; std::memcpy(&r, &out.nodes[(size_t)node * kNodeFloats + 12 + c], 4);
                if (r >= 0) st.emplace_back(r, dep + 1);
            }
        }
        out.max_depth = md;
    }
    return hipSuccess;
}

}  // namespace crt
