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
#include "crt_math.h"

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
                                                     unsigned int *__restrict__ flag, float *__restrict__ nb, float *__restrict__ nodes,
                                                     int *__restrict__ height)
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
        if (height) {                                       // inner nodes on the longest way down (the root's = the tree's depth)
            const int h0 = c0 < 0 ? 0 : __builtin_nontemporal_load(&height[c0]), h1 = c1 < 0 ? 0 : __builtin_nontemporal_load(&height[c1]);
            __builtin_nontemporal_store((h0 > h1 ? h0 : h1) + 1, &height[cur]);
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
                       d_flag.p, d_nb.p, d_nodes.p, (int *)nullptr);
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

// ---------------------------------------------------------------------------------------------------------------
// The whole build on the device (crt_build_accel(CRT_ACCEL_LBVH) for quantisable scenes): bounds of the primitives
// from their 80-byte records, Morton keys, radix sort, hierarchy, bottom-up bounds as above; then the collapse to the
// 4-wide tree, its 16-bit quantisation and the leaf-ordered primitive records, without a copy of the tree ever
// visiting the host.  Same structures and rules as the host path (crt_bvh.cpp collapse_bvh4 / quantize_bvh4,
// crt_api.cpp upload_geometry), so the kernels and the image are the same.
namespace {

struct RawPrim { uint32_t category; f3 d1, d2, d3; uint32_t emission, reflectance, material, index; };

__device__ __forceinline__ RawPrim load_raw(const unsigned char *__restrict__ raw, size_t i)
{
    const uint4 *r = (const uint4 *)(raw + i * 80);
    const uint4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4];
    RawPrim p;
    p.category = a.x;
    p.d1 = f3{bits_f(b.x), bits_f(b.y), bits_f(b.z)};
    p.d2 = f3{bits_f(c.x), bits_f(c.y), bits_f(c.z)};
    p.d3 = f3{bits_f(d.x), bits_f(d.y), bits_f(d.z)};
    p.emission = e.x; p.reflectance = e.y; p.material = e.z; p.index = e.w;
    return p;
}

// order-preserving map float -> uint for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f_ord(float f) { const uint32_t u = f_bits(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float ord_f(uint32_t u) { return bits_f((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// Conservative bounds of every primitive (the rules of upload_geometry: acceptance box + 2 * hit_pad, the region a
// patch's test really accepts, a radial term for spheres; non-finite -> never culled) and the box of the centroids.
__global__ __launch_bounds__(256) void k_lbvh_prim_bounds(const unsigned char *__restrict__ raw, uint32_t n, float pad,
                                                          float *__restrict__ lo, float *__restrict__ hi, uint32_t *__restrict__ cbox)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float c[3] = {0, 0, 0};
    bool cvalid = false;
    if (i < n) {
        const RawPrim p = load_raw(raw, i);
        const float S = pad * 131072.0f;
        f3 cs[4];
        int nc;
        if (p.category == 1u) {
            const float r = abs_(p.d2.x);
            cs[0] = f3{p.d1.x - r, p.d1.y - r, p.d1.z - r}; cs[1] = f3{p.d1.x + r, p.d1.y + r, p.d1.z + r}; nc = 2;
        } else {
            cs[0] = p.d1; cs[1] = p.d1 + p.d2; cs[2] = p.d1 + p.d3; nc = 3;
            if (p.category == 0u) { cs[3] = cs[1] + p.d3; nc = 4; }
        }
        float l[3] = {cs[0].x, cs[0].y, cs[0].z}, h[3] = {cs[0].x, cs[0].y, cs[0].z};
        for (int k = 1; k < nc; k++) {
            l[0] = fminf(l[0], cs[k].x); l[1] = fminf(l[1], cs[k].y); l[2] = fminf(l[2], cs[k].z);
            h[0] = fmaxf(h[0], cs[k].x); h[1] = fmaxf(h[1], cs[k].y); h[2] = fmaxf(h[2], cs[k].z);
        }
        float g = 2.0f * pad;
        if (p.category == 0u) {
            const double e1[3] = {p.d2.x, p.d2.y, p.d2.z}, e2[3] = {p.d3.x, p.d3.y, p.d3.z};
            const double g11 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
            const double g22 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
            const double g12 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
            const double det = g11 * g22 - g12 * g12;
            if (!(det > 1e-9 * g11 * g22)) {
                l[0] = l[1] = l[2] = -3.0e38f; h[0] = h[1] = h[2] = 3.0e38f;
            } else {
                const double P0[3] = {p.d1.x, p.d1.y, p.d1.z};
                for (int k = 0; k < 4; k++) {
                    const double a = (k & 1) ? g11 : 0.0, b = (k & 2) ? g22 : 0.0;
                    const double al = (a * g22 - b * g12) / det, be = (b * g11 - a * g12) / det;
                    for (int ax = 0; ax < 3; ax++) {
                        const double v = P0[ax] + al * e1[ax] + be * e2[ax];
                        l[ax] = fminf(l[ax], nextafterf((float)v, -INFINITY));
                        h[ax] = fmaxf(h[ax], nextafterf((float)v, INFINITY));
                    }
                }
            }
        }
        if (p.category == 1u) {
            const float r = fabsf(p.d2.x);
            g += (r > 0.0f) ? fminf(S * S * 9.5367431640625e-07f / r, S) : S;
        }
        cvalid = true;
        for (int a = 0; a < 3; a++) {
            if (!(l[a] == l[a]) || !(h[a] == h[a]) || isinf(l[a]) || isinf(h[a])) { l[a] = -3.0e38f; h[a] = 3.0e38f; }
            const float lv = l[a] - g, hv = h[a] + g;
            lo[3 * (size_t)i + a] = lv; hi[3 * (size_t)i + a] = hv;
            c[a] = 0.5f * lv + 0.5f * hv;
            if (!(c[a] > -1.0e30f && c[a] < 1.0e30f)) cvalid = false;
        }
    }
    // box of the (finite) centroids: wave reduce, then one atomic pair per wave and axis
    for (int a = 0; a < 3; a++) {
        float mn = cvalid && (c[a] > -1.0e30f && c[a] < 1.0e30f) ? c[a] : 3.0e38f, mx = cvalid && (c[a] > -1.0e30f && c[a] < 1.0e30f) ? c[a] : -3.0e38f;
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&cbox[a], f_ord(mn)); atomicMax(&cbox[3 + a], f_ord(mx)); }
    }
}

// Morton keys with the centroid box read from device memory (k_lbvh_keys takes it from the host).
__global__ __launch_bounds__(256) void k_lbvh_keys_dev(const float *__restrict__ lo, const float *__restrict__ hi, uint32_t n,
                                                       const uint32_t *__restrict__ cbox, unsigned long long *__restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float b[3], sc[3];
    for (int a = 0; a < 3; a++) {
        float l = ord_f(cbox[a]), h = ord_f(cbox[3 + a]);
        if (!(l <= h)) { l = 0.0f; h = 1.0f; }
        b[a] = l; sc[a] = 1024.0f / fmaxf(h - l, 1.0e-20f);
    }
    const float cx = 0.5f * lo[3 * (size_t)i + 0] + 0.5f * hi[3 * (size_t)i + 0];
    const float cy = 0.5f * lo[3 * (size_t)i + 1] + 0.5f * hi[3 * (size_t)i + 1];
    const float cz = 0.5f * lo[3 * (size_t)i + 2] + 0.5f * hi[3 * (size_t)i + 2];
    const float fx = (cx - b[0]) * sc[0], fy = (cy - b[1]) * sc[1], fz = (cz - b[2]) * sc[2];
    const uint32_t qx = fx > 0.0f ? (fx < 1023.0f ? (uint32_t)fx : 1023u) : 0u;
    const uint32_t qy = fy > 0.0f ? (fy < 1023.0f ? (uint32_t)fy : 1023u) : 0u;
    const uint32_t qz = fz > 0.0f ? (fz < 1023.0f ? (uint32_t)fz : 1023u) : 0u;
    const uint32_t m = (expand10(qx) << 2) | (expand10(qy) << 1) | expand10(qz);
    keys[i] = ((unsigned long long)m << 32) | i;
}

struct CBox { float lo[3], hi[3]; int ref; };
__device__ __forceinline__ void load_children(const float *__restrict__ nodes2, int node, CBox &a, CBox &b)
{
    const float4 *nd = (const float4 *)(nodes2 + (size_t)node * kNodeFloats);
    const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
    a.lo[0] = n0.x; a.lo[1] = n0.y; a.lo[2] = n0.z; a.hi[0] = n0.w; a.hi[1] = n1.x; a.hi[2] = n1.y;
    b.lo[0] = n1.z; b.lo[1] = n1.w; b.lo[2] = n2.x; b.hi[0] = n2.y; b.hi[1] = n2.z; b.hi[2] = n2.w;
    a.ref = (int)f_bits(n3.x); b.ref = (int)f_bits(n3.y);
}
__device__ __forceinline__ float cbox_area(const CBox &c)
{
    const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// One level of the collapse: thread t turns BVH2 node frontier_in[t] into the 4-wide node level_base + t (open the
// inner child with the largest surface until there are four: collapse_bvh4's rule), quantises its child boxes
// (quantize_bvh4's rule: 16-bit grid over the scene box, rounded outward with one grid unit of slack) and appends the
// children that are inner nodes to the next level's frontier; their 4-wide ids are next_base + position.
__global__ __launch_bounds__(256) void k_lbvh_collapse_level(const float *__restrict__ nodes2, const int *__restrict__ frontier_in, uint32_t count,
                                                             uint32_t level_base, int *__restrict__ frontier_out, uint32_t *__restrict__ out_count,
                                                             uint32_t next_base, uint4 *__restrict__ nodes4q,
                                                             double bx, double by, double bz, double sx, double sy, double sz)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    CBox ch[4];
    int n = 2;
    load_children(nodes2, frontier_in[t], ch[0], ch[1]);
    while (n < 4) {
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < n; i++) if (ch[i].ref >= 0) { const float a = cbox_area(ch[i]); if (a > ba) { ba = a; best = i; } }
        if (best < 0) break;
        CBox a, b;
        load_children(nodes2, ch[best].ref, a, b);
        ch[best] = a; ch[n++] = b;
    }
    uint32_t k = 0;
    for (int i = 0; i < n; i++) k += ch[i].ref >= 0 ? 1u : 0u;
    uint32_t pos = 0;
    if (k) pos = atomicAdd(out_count, k);
    int refs[4] = {0, 0, 0, 0};
    uint32_t j = 0;
    for (int i = 0; i < n; i++) {
        if (ch[i].ref >= 0) { frontier_out[pos + j] = ch[i].ref; refs[i] = (int)(next_base + pos + j); j++; }
        else refs[i] = ch[i].ref;
    }
    const double base[3] = {bx, by, bz}, scale[3] = {sx, sy, sz};
    uint32_t q[24];
    for (int i = 0; i < 4; i++)
        for (int a = 0; a < 3; a++) {
            if (i >= n) { q[4 * a + i] = 65535u; q[12 + 4 * a + i] = 0u; continue; }      // empty slot: never hit
            const double l = ((double)ch[i].lo[a] - base[a]) / scale[a], h = ((double)ch[i].hi[a] - base[a]) / scale[a];
            long long ql = (long long)floor(l) - 1, qh = (long long)ceil(h) + 1;
            ql = ql < 0 ? 0 : (ql > 65535 ? 65535 : ql); qh = qh < 0 ? 0 : (qh > 65535 ? 65535 : qh);
            q[4 * a + i] = (uint32_t)ql; q[12 + 4 * a + i] = (uint32_t)qh;
        }
    uint4 *o = nodes4q + 4 * (size_t)(level_base + t);
    o[0] = uint4{q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), q[6] | (q[7] << 16)};
    o[1] = uint4{q[8] | (q[9] << 16), q[10] | (q[11] << 16), q[12] | (q[13] << 16), q[14] | (q[15] << 16)};
    o[2] = uint4{q[16] | (q[17] << 16), q[18] | (q[19] << 16), q[20] | (q[21] << 16), q[22] | (q[23] << 16)};
    o[3] = uint4{(uint32_t)refs[0], (uint32_t)refs[1], (uint32_t)refs[2], (uint32_t)refs[3]};
}

// Leaf-ordered primitive records (crt_device.h: A, B, C per slot, D for patches) and the inverse permutation.
__global__ __launch_bounds__(256) void k_lbvh_gather_prims(const unsigned char *__restrict__ raw, const unsigned long long *__restrict__ keys, uint32_t n,
                                                           float4 *__restrict__ prim, float4 *__restrict__ primD, uint32_t *__restrict__ slot_of_index)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= n) return;
    const RawPrim p = load_raw(raw, (size_t)(keys[slot] & 0xFFFFFFFFull));
    const uint32_t meta = (p.category & 3u) | ((p.material & 3u) << 2) | ((p.emission & 0x3FFFu) << 4) | ((p.reflectance & 0x3FFFu) << 18);
    float4 A = {p.d1.x, p.d1.y, p.d1.z, bits_f(meta)};
    float4 B = {p.d2.x, p.d2.y, p.d2.z, bits_f(p.index)};
    float4 C = {p.d3.x, p.d3.y, p.d3.z, 0.0f};
    float4 D = {0.0f, 0.0f, 0.0f, 0.0f};
    if (p.category == 0u) {
        const f3 nrm = normalize(cross(p.d2, p.d3));             // ComputeShader.wgsl:536
        D = float4{nrm.x, nrm.y, nrm.z, dot(p.d2, p.d2)};        // :563 denominator
        C.w = dot(p.d3, p.d3);                                   // :564 denominator
    } else if (p.category == 1u) {
        const float r = p.d2.x;                                  // :593-594
        B = float4{r, r * r, 0.0f, bits_f(p.index)};
    }
    prim[3 * (size_t)slot + 0] = A; prim[3 * (size_t)slot + 1] = B; prim[3 * (size_t)slot + 2] = C;
    primD[slot] = D;
    slot_of_index[p.index] = slot;
}

}  // namespace

// d_raw: the scene's 80-byte records on the device.  Outputs (device, caller-allocated): prim 3n float4, primD n,
// slot_of_index n, nodes2 (n-1) x 16 floats (the BVH2 of crt_bvh.h), nodes4q (n-1) x 4 uint4 at most.
// res.quantised == false: the scene cannot be quantised (same rules as quantize_bvh4); nothing usable was produced
// beyond nodes2 and the caller takes the host path.
hipError_t build_lbvh_device(const unsigned char *d_raw, uint32_t n, float hit_pad, float4 *d_prim, float4 *d_primD,
                             uint32_t *d_slot_of_index, float *d_nodes2, uint4 *d_nodes4q, LbvhDeviceResult &res, hipStream_t stream)
{
    res = LbvhDeviceResult();
    if (n < 2) return hipErrorInvalidValue;
    Tmp<float> d_lo, d_hi, d_nb;
    Tmp<unsigned long long> d_keys, d_sorted;
    Tmp<int> d_child, d_pi, d_pl, d_height, d_front[2];
    Tmp<unsigned int> d_flag, d_small;
    Tmp<char> d_tmp;
    hipError_t e;
#define LB(call) do { e = (call); if (e != hipSuccess) return e; } while (0)
    LB(d_lo.alloc((size_t)n * 3)); LB(d_hi.alloc((size_t)n * 3));
    LB(d_keys.alloc(n)); LB(d_sorted.alloc(n));
    LB(d_child.alloc((size_t)2 * (n - 1))); LB(d_pi.alloc(n - 1)); LB(d_pl.alloc(n));
    LB(d_flag.alloc(n - 1)); LB(d_nb.alloc((size_t)6 * (n - 1))); LB(d_height.alloc(n - 1));
    LB(d_small.alloc(16));
    const uint32_t cinit[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    LB(hipMemcpyAsync(d_small.p, cinit, sizeof cinit, hipMemcpyHostToDevice, stream));
    LB(hipMemsetAsync(d_flag.p, 0, (size_t)(n - 1) * 4, stream));
    const unsigned blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_lbvh_prim_bounds, dim3(blocks), dim3(256), 0, stream, d_raw, n, hit_pad, d_lo.p, d_hi.p, d_small.p);
    LB(hipGetLastError());
    hipLaunchKernelGGL(k_lbvh_keys_dev, dim3(blocks), dim3(256), 0, stream, d_lo.p, d_hi.p, n, d_small.p, d_keys.p);
    LB(hipGetLastError());
    size_t tmp_bytes = 0;
    LB(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys.p, d_sorted.p, (int)n, 0, 62, stream));
    LB(d_tmp.alloc(tmp_bytes));
    LB(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tmp_bytes, d_keys.p, d_sorted.p, (int)n, 0, 62, stream));
    hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(blocks), dim3(256), 0, stream, d_sorted.p, (int)n, d_child.p, d_pi.p, d_pl.p);
    LB(hipGetLastError());
    hipLaunchKernelGGL(k_lbvh_bounds, dim3(blocks), dim3(256), 0, stream, d_sorted.p, (int)n, d_lo.p, d_hi.p, d_child.p, d_pi.p, d_pl.p,
                       d_flag.p, d_nb.p, d_nodes2, d_height.p);
    LB(hipGetLastError());
    // the leaf-ordered primitive records do not depend on the tree's shape: enqueue them now
    hipLaunchKernelGGL(k_lbvh_gather_prims, dim3(blocks), dim3(256), 0, stream, d_raw, d_sorted.p, n, d_prim, d_primD, d_slot_of_index);
    LB(hipGetLastError());
    // the scene box = the union of the root's two child boxes; the tree's depth = the root's height
    float rootrec[kNodeFloats];
    int depth = 0;
    LB(hipMemcpyAsync(rootrec, d_nodes2, sizeof rootrec, hipMemcpyDeviceToHost, stream));
    LB(hipMemcpyAsync(&depth, d_height.p, sizeof depth, hipMemcpyDeviceToHost, stream));
    LB(hipStreamSynchronize(stream));
    res.max_depth = (uint32_t)depth;
    double base[3], scale[3];
    for (int a = 0; a < 3; a++) {
        const float glo = std::min(rootrec[a], rootrec[6 + a]), ghi = std::max(rootrec[3 + a], rootrec[9 + a]);
        if (!(glo > -1.0e30f) || !(ghi < 1.0e30f)) return hipSuccess;          // unbounded primitive: not quantisable
        const float ext = std::max(ghi - glo, 1.0e-3f);
        const float mag = std::max(std::fabs(glo), std::fabs(ghi));
        if (mag > 16.0f * ext) return hipSuccess;                               // too far from the origin for the slack
        res.qbase[a] = glo; res.qscale[a] = ext / 65533.0f;
        base[a] = res.qbase[a]; scale[a] = res.qscale[a];
    }
    // collapse level by level (breadth-first numbering: the top of the tree sits together, like the host's renumbering)
    LB(d_front[0].alloc(n / 2 + 2)); LB(d_front[1].alloc(n / 2 + 2));
    const int root = 0;
    LB(hipMemcpyAsync(d_front[0].p, &root, sizeof root, hipMemcpyHostToDevice, stream));
    uint32_t count = 1, level_base = 0;
    int cur = 0;
    for (int level = 0; count > 0; level++) {
        if (level > 64) return hipErrorUnknown;
        LB(hipMemsetAsync(d_small.p + 8, 0, 4, stream));
        hipLaunchKernelGGL(k_lbvh_collapse_level, dim3((count + 255u) / 256u), dim3(256), 0, stream, (const float *)d_nodes2, (const int *)d_front[cur].p, count,
                           level_base, d_front[cur ^ 1].p, d_small.p + 8, level_base + count, d_nodes4q, base[0], base[1], base[2], scale[0], scale[1], scale[2]);
        LB(hipGetLastError());
        uint32_t next = 0;
        LB(hipMemcpyAsync(&next, d_small.p + 8, 4, hipMemcpyDeviceToHost, stream));
        LB(hipStreamSynchronize(stream));
        level_base += count;
        count = next;
        cur ^= 1;
    }
#undef LB
    res.n_nodes4 = level_base;
    res.quantised = true;
    return hipSuccess;
}

}  // namespace crt