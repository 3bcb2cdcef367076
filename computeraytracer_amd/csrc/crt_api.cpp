// crt_api.cpp -- the C ABI of include/crt.h on top of the gfx950 kernels.
//
// Host-side responsibilities (the reference does these in src/main.js):
//   upload   main.js:147-393  -> crt_upload_scene (80-byte records -> device layout)
//   state    main.js:298-311  -> accumulator + sample counter owned by the context
//   dispatch main.js:597-611  -> crt_trace(n) == n x {sample++ ; trace}
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/crt.h"
#include "crt_bvh.h"
#include "crt_device.h"
#include "crt_math.h"

namespace crt {
hipError_t launch_trace(const TraceParams &P, bool count, bool brute, hipStream_t stream);
hipError_t launch_debug_intersect(const DevScene &S, const float *rays, size_t n, float *out, int brute,
                                  hipStream_t stream);
hipError_t launch_debug_math(int fn, const float *a, const float *b, float *out, size_t n, hipStream_t stream);
hipError_t wf_launch_init(const WfParams &P, hipStream_t s);
hipError_t wf_launch_tea(const WfParams &P, uint32_t *out, hipStream_t s);
hipError_t wf_launch_shade(const WfParams &P, uint32_t it, hipStream_t s);
hipError_t wf_launch_trace(const WfParams &P, uint32_t it, uint32_t trace_blocks, hipStream_t s);
hipError_t wf_launch_finish(const WfParams &P, uint32_t max_paths, hipStream_t s);
hipError_t wf_launch_resolve(const WfParams &P, uint32_t last_sample, hipStream_t s);
hipError_t build_lbvh(const float *lo, const float *hi, uint32_t n, Bvh &out, hipStream_t stream);
}  // namespace crt

using namespace crt;

namespace {

std::string g_create_error;

// Host copy of one 80-byte record (ComputeShader.wgsl:41-47, main.js:211-246).
struct HostPrim {
    uint32_t category;
    f3 d1, d2, d3;
    uint32_t emission, reflectance, material, index;
};

HostPrim read_prim(const uint8_t *base, size_t i)
{
    HostPrim p;
    const uint8_t *r = base + i * 80;
    float f[9];
    uint32_t u[4];
    std::memcpy(&p.category, r, 4);
    std::memcpy(f, r + 16, 12); std::memcpy(f + 3, r + 32, 12); std::memcpy(f + 6, r + 48, 12);
    std::memcpy(u, r + 64, 16);
    p.d1 = f3{f[0], f[1], f[2]}; p.d2 = f3{f[3], f[4], f[5]}; p.d3 = f3{f[6], f[7], f[8]};
    p.emission = u[0]; p.reflectance = u[1]; p.material = u[2]; p.index = u[3];
    return p;
}

struct WfRun;
int wf_flush(struct ::crt_ctx *c);
int wf_check_dropped(struct ::crt_ctx *c);

// Test hook (option "debug_fail_alloc" = k): the k-th device allocation from now on reports out-of-memory.
long long g_fail_alloc_in = 0;

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        if (count == 0) return hipSuccess;
        const bool inject = g_fail_alloc_in > 0 && --g_fail_alloc_in == 0;
        const hipError_t e = inject ? hipErrorOutOfMemory : hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }   // n stays 0: a later "is it large enough" test re-allocates
        n = count;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct crt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    // host copies
    std::vector<HostPrim> prims;
    std::vector<HostPrim> lights;
    float camera[16] = {0};
    uint32_t W = 0, H = 0;
    bool have_scene = false;
    int accel_mode = -1;            // -1: not built
    bool want_lbvh = false;         // crt_build_accel(CRT_ACCEL_LBVH): build the BVH2 on the GPU
    int accel_builder = 0;          // 0: host binned SAH, 1: GPU LBVH
    Bvh bvh;
    Bvh4 bvh4;
    Bvh4Q bvh4q;
    Bvh8Q bvh8q;
    int quantize = 1;
    int wf_width = 4;               // node width of the wavefront traversal: 4 (64-byte quantised nodes), or 8 (128-byte; measured slower)

    // device scene
    DevBuf<float4> d_prim, d_primD, d_nodes, d_nodes4, d_lights;
    DevBuf<int> w_overflow;
    DevBuf<uint4> d_nodes4q, d_nodes8q;
    DevBuf<uint32_t> d_slot_of_index;
    DevBuf<float> d_spectra, d_cie;
    DevScene sc{};

    // tile + outputs
    uint32_t x0 = 0, y0 = 0, tw = 0, th = 0;
    uint32_t band = 0x40000000u, stride = 1, phase = 0;   // row interleave (rectangular tile by default)
    DevBuf<float4> d_accum;
    DevBuf<uchar4> d_rgba;
    float4 *accum_bound = nullptr;
    uchar4 *rgba_bound = nullptr;
    uint32_t sample = 0;

    DevBuf<unsigned long long> d_counters;
    bool counting = false;
    float last_ms = 0.0f;
    uint32_t last_launches = 0;
    bool last_timed = false;
    uint32_t spp_per_launch = 0;    // 0 = auto

    // wavefront pipeline (crt_wavefront.hip)
    int pipeline = 1;               // 1 = wavefront (default), 0 = v1 megakernel
    uint32_t wf_pool = 0;           // 0 = auto
    uint32_t wf_waves_per_cu = 16;  // per pipe
    int num_cu = 0;
    DevBuf<float4> w_ray_o, w_ray_d, w_sh_d, w_beta, w_radiance, w_nee, w_staging[kWfRing];
    DevBuf<uint4> w_rng, w_misc;
    DevBuf<float2> w_hit;
    DevBuf<uint32_t> w_vis, w_list_ext, w_tea;
    // up to kMaxPipes half-pools, each its own shade->trace chain on its own stream
    static constexpr int kMaxPipes = 4;
    int wf_pipes = 2;
    int wf_defer = 1;               // 1: a batch ends parked, its last paths finish under the next batch (or at crt_sync)
    int wf_tail_walk = 1;           // shade walks the ray lists once few paths are left
    int wf_park_its = 4;            // a batch parks when its queue holds less than this many iterations' worth
    int wf_chunk = 2;               // iterations enqueued per status readback (the host's decisions lag by two chunks)
    int wf_ring = 4;                // batches in flight at most (2..kWfRing): bounds how many calls a bound output can lag
    WfRun *run = nullptr;           // pipeline state between calls
    uint32_t wf_finish_at = 32768;  // paths of the oldest batch left (per pipe) at which they move to the side pool; 0 = never
    uint32_t wf_flush_at = 4096;    // the same for the LAST batch at crt_sync (nothing to hide its tail under); 0 = never
    uint32_t wf_side_ppw = 64, wf_flush_ppw = 4;   // k_wf_finish: paths per wave, under the next batch / at crt_sync
    DevBuf<WfCtl> w_ctl[kMaxPipes];
    DevBuf<WfWorkQ> w_wq;
    WfWorkQ *h_wq[kMaxPipes][2] = {};                      // pinned: each pipe's snapshots of the two work queues
    WfCtl *h_ctl[kMaxPipes][2] = {};                       // pinned, double-buffered status readbacks
    uint32_t *h_dropped = nullptr;                         // pinned [kMaxPipes]: WfCtl::dropped after the last flush
    hipEvent_t ev_ctl[kMaxPipes][2] = {};
    hipStream_t pipe_stream[kMaxPipes] = {};               // the pipes' own streams (the context's stream only forks and resolves)
    hipEvent_t ev_fork = nullptr, ev_join[kMaxPipes] = {}, ev_evict[kMaxPipes] = {};
    bool time_kernels = false;
    std::vector<hipEvent_t> kev;    // event pairs around k_wf_trace launches
    float last_trace_kernel_ms = 0.0f;
    uint32_t last_trace_kernel_launches = 0;
    uint32_t last_iterations = 0;
    unsigned long long probes[8] = {0};   // traversal-efficiency probes of the counting kernels
};

namespace {

int fail(crt_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(c, e_ == hipErrorOutOfMemory ? CRT_ENOMEM : CRT_EDEVICE, "%s: %s", #call, \
                        hipGetErrorString(e_));                                                   \
    } while (0)

float4 *accum_ptr(crt_ctx *c) { return c->accum_bound ? c->accum_bound : c->d_accum.p; }
uchar4 *rgba_ptr(crt_ctx *c) { return c->rgba_bound ? c->rgba_bound : c->d_rgba.p; }

int alloc_tile(crt_ctx *c)
{
    size_t n = (size_t)c->tw * c->th;
    HIPCHK(c, c->d_accum.alloc(n));
    HIPCHK(c, c->d_rgba.alloc(n));
    return CRT_OK;
}

int zero_state(crt_ctx *c)
{
    size_t n = (size_t)c->tw * c->th;
    if (n) {
        HIPCHK(c, hipMemsetAsync(accum_ptr(c), 0, n * sizeof(float4), c->stream));
        HIPCHK(c, hipMemsetAsync(rgba_ptr(c), 0, n * sizeof(uchar4), c->stream));
    }
    c->sample = 0;
    return CRT_OK;
}

// Primitive corners for bounds / scene scale (same op order as the oracle's orc_hit_pad).
int prim_corners(const HostPrim &p, f3 out[4])
{
    if (p.category == 1u) {
        float r = abs_(p.d2.x);
        out[0] = f3{p.d1.x - r, p.d1.y - r, p.d1.z - r};
        out[1] = f3{p.d1.x + r, p.d1.y + r, p.d1.z + r};
        return 2;
    }
    out[0] = p.d1; out[1] = p.d1 + p.d2; out[2] = p.d1 + p.d3;
    if (p.category == 0u) { out[3] = out[1] + p.d3; return 4; }
    return 3;
}

float scene_hit_pad(const std::vector<HostPrim> &prims, const float cam[16])
{
    float S = 0.0f;
    f3 c[4];
    for (const HostPrim &p : prims) {
        int nc = prim_corners(p, c);
        for (int k = 0; k < nc; k++) {
            S = max_(S, abs_(c[k].x)); S = max_(S, abs_(c[k].y)); S = max_(S, abs_(c[k].z));
        }
    }
    S = max_(S, abs_(cam[0])); S = max_(S, abs_(cam[1])); S = max_(S, abs_(cam[2]));
    return S * 7.62939453125e-06f;  // 2^-17
}

// ComputeShader.wgsl:470-487, everything independent of the pixel.
void camera_frame(const float cam[16], float out[12])
{
    f3 eye = f3{cam[0], cam[1], cam[2]}, lookat = f3{cam[4], cam[5], cam[6]}, up = f3{cam[8], cam[9], cam[10]};
    f3 w = normalize(eye - lookat);
    f3 u = normalize(cross(up, w));
    f3 v = cross(w, u);
    float aspect_ratio = cam[11] / cam[12];
    float viewport_height = 2.0f * tan_(cam[13] / 2.0f);
    float viewport_width = aspect_ratio * viewport_height;
    f3 horizontal = u * viewport_width;
    f3 vertical = v * viewport_height;
    f3 llc = ((eye - horizontal / 2.0f) - vertical / 2.0f) - w;
    out[0] = llc.x; out[1] = llc.y; out[2] = llc.z;
    out[3] = horizontal.x; out[4] = horizontal.y; out[5] = horizontal.z;
    out[6] = vertical.x; out[7] = vertical.y; out[8] = vertical.z;
    out[9] = eye.x; out[10] = eye.y; out[11] = eye.z;
}

// Builds the device primitive arrays in `order` and (for BVH2) the node array.
int upload_geometry(crt_ctx *c, int mode)
{
    const uint32_t n = (uint32_t)c->prims.size();
    const float pad = c->sc.hit_pad;
    std::vector<uint32_t> order;
    c->bvh = Bvh();
    if (mode == CRT_ACCEL_BVH2 && n > 0) {
        // Conservative boxes: the triangle acceptance box is [corner min - pad, corner max + pad];
        // node boxes get 2*pad (covers the slab arithmetic), spheres an extra radial term.
        float S = pad * 131072.0f;
        std::vector<float> lo((size_t)n * 3), hi((size_t)n * 3);
        f3 cs[4];
        for (uint32_t i = 0; i < n; i++) {
            const HostPrim &p = c->prims[i];
            int nc = prim_corners(p, cs);
            float l[3] = {cs[0].x, cs[0].y, cs[0].z}, h[3] = {cs[0].x, cs[0].y, cs[0].z};
            for (int k = 1; k < nc; k++) {
                l[0] = std::min(l[0], cs[k].x); l[1] = std::min(l[1], cs[k].y); l[2] = std::min(l[2], cs[k].z);
                h[0] = std::max(h[0], cs[k].x); h[1] = std::max(h[1], cs[k].y); h[2] = std::max(h[2], cs[k].z);
            }
            float g = 2.0f * pad;
            if (p.category == 0u) {
                // The patch test accepts {P0+m : 0<=m.e1<=e1.e1, 0<=m.e2<=e2.e2} (ComputeShader.wgsl
                // :563-566 use projections, which only equals the corner parallelogram when e1 is
                // perpendicular to e2 -- cornell's box faces are not).  Bound THAT region.
                double e1[3] = {p.d2.x, p.d2.y, p.d2.z}, e2[3] = {p.d3.x, p.d3.y, p.d3.z};
                double g11 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
                double g22 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
                double g12 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
                double det = g11 * g22 - g12 * g12;
                if (!(det > 1e-9 * g11 * g22)) {
                    l[0] = l[1] = l[2] = -3.0e38f; h[0] = h[1] = h[2] = 3.0e38f;   // unbounded strip
                } else {
                    double P0[3] = {p.d1.x, p.d1.y, p.d1.z};
                    for (int k = 0; k < 4; k++) {
                        double a = (k & 1) ? g11 : 0.0, b = (k & 2) ? g22 : 0.0;
                        double al = (a * g22 - b * g12) / det, be = (b * g11 - a * g12) / det;
                        for (int ax = 0; ax < 3; ax++) {
                            double v = P0[ax] + al * e1[ax] + be * e2[ax];
                            l[ax] = std::min(l[ax], (float)std::nextafter((float)v, -INFINITY));
                            h[ax] = std::max(h[ax], (float)std::nextafter((float)v, INFINITY));
                        }
                    }
                }
            }
            if (p.category == 1u) {
                float r = std::fabs(p.d2.x);
                g += (r > 0.0f) ? std::min(S * S * 9.5367431640625e-07f / r, S) : S;
            }
            for (int a = 0; a < 3; a++) {
                if (!(l[a] == l[a]) || !(h[a] == h[a]) || std::isinf(l[a]) || std::isinf(h[a])) {
                    l[a] = -3.0e38f; h[a] = 3.0e38f;      // non-finite primitive: never culled
                }
                lo[3 * i + a] = l[a] - g; hi[3 * i + a] = h[a] + g;
            }
        }
        c->accel_builder = 0;
        if (c->want_lbvh && n >= 2) {
            // GPU build (crt_lbvh.hip): same structure, so everything below is shared
            hipError_t e = build_lbvh(lo.data(), hi.data(), n, c->bvh, c->stream);
            if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? CRT_ENOMEM : CRT_EDEVICE, "crt_build_accel: GPU LBVH build: %s", hipGetErrorString(e));
            c->accel_builder = 1;
        } else {
            build_bvh2(lo.data(), hi.data(), n, c->bvh);
        }
        order = c->bvh.order;
    } else {
        order.resize(n);
        for (uint32_t i = 0; i < n; i++) order[i] = i;
        c->bvh.root = -1;
    }

    std::vector<float4> hp((size_t)n * 3), hd(n);
    std::vector<uint32_t> slot_of(n);
    for (uint32_t slot = 0; slot < n; slot++) {
        const HostPrim &p = c->prims[order[slot]];
        slot_of[p.index] = slot;
        uint32_t meta = (p.category & 3u) | ((p.material & 3u) << 2) | ((p.emission & 0x3FFFu) << 4) |
                        ((p.reflectance & 0x3FFFu) << 18);
        float4 A = {p.d1.x, p.d1.y, p.d1.z, bits_f(meta)};
        float4 B = {p.d2.x, p.d2.y, p.d2.z, bits_f(p.index)};
        float4 C = {p.d3.x, p.d3.y, p.d3.z, 0.0f};
        float4 D = {0.0f, 0.0f, 0.0f, 0.0f};
        if (p.category == 0u) {
            f3 nrm = normalize(cross(p.d2, p.d3));               // ComputeShader.wgsl:536
            D = float4{nrm.x, nrm.y, nrm.z, dot(p.d2, p.d2)};    // :563 denominator
            C.w = dot(p.d3, p.d3);                               // :564 denominator
        } else if (p.category == 1u) {
            float r = p.d2.x;                                    // :593-594
            B = float4{r, r * r, 0.0f, bits_f(p.index)};
        }
        hp[3 * (size_t)slot + 0] = A; hp[3 * (size_t)slot + 1] = B; hp[3 * (size_t)slot + 2] = C;
        hd[slot] = D;
    }
    HIPCHK(c, c->d_prim.alloc(std::max<size_t>(hp.size(), 3)));
    HIPCHK(c, c->d_primD.alloc(std::max<size_t>(hd.size(), 1)));
    HIPCHK(c, c->d_slot_of_index.alloc(std::max<size_t>(n, 1)));
    if (n) {
        HIPCHK(c, hipMemcpy(c->d_prim.p, hp.data(), hp.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_primD.p, hd.data(), hd.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_slot_of_index.p, slot_of.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    size_t nn = c->bvh.nodes.size() / 4;
    HIPCHK(c, c->d_nodes.alloc(std::max<size_t>(nn, 4)));
    if (nn) HIPCHK(c, hipMemcpy(c->d_nodes.p, c->bvh.nodes.data(), nn * sizeof(float4), hipMemcpyHostToDevice));
    c->sc.prim = c->d_prim.p;
    c->sc.primD = c->d_primD.p;
    c->sc.slot_of_index = c->d_slot_of_index.p;
    c->bvh4 = Bvh4();
    if (mode == CRT_ACCEL_BVH2 && n > 0) collapse_bvh4(c->bvh, c->bvh4);
    size_t nn4 = c->bvh4.nodes.size() / 4;
    HIPCHK(c, c->d_nodes4.alloc(std::max<size_t>(nn4, 8)));
    if (nn4) HIPCHK(c, hipMemcpy(c->d_nodes4.p, c->bvh4.nodes.data(), nn4 * sizeof(float4), hipMemcpyHostToDevice));
    c->sc.nodes4 = c->d_nodes4.p;
    c->sc.root4 = c->bvh4.root;
    c->sc.n_nodes4 = c->bvh4.n_inner;
    c->bvh4q = Bvh4Q();
    c->sc.nodes4q = nullptr;
    if (c->quantize && c->bvh4.n_inner) quantize_bvh4(c->bvh4, c->bvh4q);
    if (c->bvh4q.ok) {
        size_t nq = c->bvh4q.nodes.size() / 4;
        HIPCHK(c, c->d_nodes4q.alloc(nq));
        HIPCHK(c, hipMemcpy(c->d_nodes4q.p, c->bvh4q.nodes.data(), nq * sizeof(uint4), hipMemcpyHostToDevice));
        c->sc.nodes4q = c->d_nodes4q.p;
        for (int a = 0; a < 3; a++) { c->sc.qbase[a] = c->bvh4q.base[a]; c->sc.qscale[a] = c->bvh4q.scale[a]; }
    }
    c->bvh8q = Bvh8Q();
    c->sc.nodes8q = nullptr;
    c->sc.root8 = -1;
    if (c->quantize && c->wf_width == 8 && c->bvh4q.ok) build_bvh8q(c->bvh, c->bvh8q);
    if (c->bvh8q.ok) {
        const size_t nq = c->bvh8q.nodes.size() / 4;
        HIPCHK(c, c->d_nodes8q.alloc(nq));
        HIPCHK(c, hipMemcpy(c->d_nodes8q.p, c->bvh8q.nodes.data(), nq * sizeof(uint4), hipMemcpyHostToDevice));
        c->sc.nodes8q = c->d_nodes8q.p;
        c->sc.root8 = c->bvh8q.root;
        for (int a = 0; a < 3; a++) { c->sc.qbase[a] = c->bvh8q.base[a]; c->sc.qscale[a] = c->bvh8q.scale[a]; }
    }
    c->sc.nodes = c->d_nodes.p;
    c->sc.root = c->bvh.root;
    c->sc.nprim = n;
    c->accel_mode = mode;
    return CRT_OK;
}


// ---------------------------------------------------------------- wavefront driver
int wf_ensure(crt_ctx *c, size_t P, size_t staging_elems, size_t list_elems)
{
    if (c->w_list_ext.n < list_elems) HIPCHK(c, c->w_list_ext.alloc(list_elems));
    // (each array on its own: after a failed allocation that array reports n == 0 and is retried by the next call)
    if (c->w_ray_o.n < P) HIPCHK(c, c->w_ray_o.alloc(P));
    if (c->w_ray_d.n < P) HIPCHK(c, c->w_ray_d.alloc(P));
    if (c->w_sh_d.n < P) HIPCHK(c, c->w_sh_d.alloc(P));
    if (c->w_beta.n < P) HIPCHK(c, c->w_beta.alloc(P));
    if (c->w_radiance.n < P) HIPCHK(c, c->w_radiance.alloc(P));
    if (c->w_nee.n < P) HIPCHK(c, c->w_nee.alloc(P));
    if (c->w_rng.n < P) HIPCHK(c, c->w_rng.alloc(P));
    if (c->w_misc.n < P) HIPCHK(c, c->w_misc.alloc(P));
    if (c->w_hit.n < P) HIPCHK(c, c->w_hit.alloc(P));
    if (c->w_vis.n < P) HIPCHK(c, c->w_vis.alloc(P));
    for (uint32_t b = 0; b < kWfRing; b++)
        if (c->w_staging[b].n < staging_elems) HIPCHK(c, c->w_staging[b].alloc(staging_elems));
    if (c->w_tea.n < (size_t)c->tw * c->th) HIPCHK(c, c->w_tea.alloc((size_t)c->tw * c->th));
    if (!c->w_wq.p) {
        HIPCHK(c, c->w_wq.alloc(kWfRing));
        HIPCHK(c, hipMemset(c->w_wq.p, 0, kWfRing * sizeof(WfWorkQ)));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(c, hipHostMalloc((void **)&c->h_dropped, crt_ctx::kMaxPipes * sizeof(uint32_t), hipHostMallocDefault));
        std::memset(c->h_dropped, 0, crt_ctx::kMaxPipes * sizeof(uint32_t));
        for (int p = 0; p < crt_ctx::kMaxPipes; p++) {
            HIPCHK(c, c->w_ctl[p].alloc(1));
            HIPCHK(c, hipMemset(c->w_ctl[p].p, 0, sizeof(WfCtl)));
            for (int b = 0; b < 2; b++) {
                HIPCHK(c, hipHostMalloc((void **)&c->h_ctl[p][b], sizeof(WfCtl), hipHostMallocDefault));
                HIPCHK(c, hipEventCreateWithFlags(&c->ev_ctl[p][b], hipEventDisableTiming));
            }
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[p], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_evict[p], hipEventDisableTiming));
        }
        for (int p = 0; p < crt_ctx::kMaxPipes; p++)
            for (int b = 0; b < 2; b++) HIPCHK(c, hipHostMalloc((void **)&c->h_wq[p][b], kWfRing * sizeof(WfWorkQ), hipHostMallocDefault));
    }
    if (c->num_cu == 0) {
        hipDeviceProp_t prop;
        HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
        c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    {   // deep-stack overflow area: 64 levels beyond the LDS part for every resident traversal lane
        const size_t lanes = (size_t)c->num_cu * c->wf_waves_per_cu * 64u * (size_t)std::max(1, c->wf_pipes);
        if (c->w_overflow.n < lanes * 64) HIPCHK(c, c->w_overflow.alloc(lanes * 64));
    }
    return CRT_OK;
}

// One shade->trace chain over its share of the pool.
struct WfPipe {
    WfParams W{};
    hipStream_t stream = nullptr;
    uint32_t it = 0, chunk = 2, tail_bound = 0, blocks_now = 0, it_end[2] = {0, 0};
    int cur = 0;                    // status buffer of the chunk that is outstanding between driver passes
    bool done = false;              // no more chunks are enqueued for this pipe (drained, evicted or parked)
    bool any = false;               // a status has been read since the newest batch began
    unsigned long long rays = 0;    // from the last status read: rays listed
    unsigned long long old = 0;     // ... and paths of the OLDEST unresolved batch still in this pipe's pool
    bool old_valid = false;         // ... counted by a launch that already knew which batch is the oldest
    uint32_t old_from = 0;          // iterations >= old_from count survivors of the current oldest batch
    bool dry[kWfRing] = {};         // a status of this pipe saw that batch's queue empty (its OWN snapshot: its
                                    // survivor count only means something once no more such paths can start here)
    uint32_t evict_next = 0;        // evict_mask for the first shade launch of the next chunk
    uint32_t it_confirmed = 0;      // iterations < it_confirmed are known to have completed (a status of them was read)
};

// A batch of samples whose paths are (or may still be) in flight.
struct WfBatch {
    uint32_t n = 0, last_sample = 0, id = 0;
    uint32_t from_it[crt_ctx::kMaxPipes] = {};   // per pipe: statuses of iterations >= from_it know this batch's queue
};

// The pipeline's state between driver calls.  Up to kWfRing batches are in flight, each with its own work
// queue, staging buffer and side pools (indexed by the batch id, which a path carries in its flags).  A batch
// normally ENDS PARKED: crt_trace returns while its queue still holds a few iterations' worth of work and one
// chunk of iterations is enqueued; the next call publishes its queue and dead slots re-arm from the oldest
// non-empty queue on -- the pool never runs dry between calls, however small the calls are.  Once only a few
// paths of the OLDEST batch are left, k_wf_shade moves them to the side pool, k_wf_finish runs them to their
// end and the batch is resolved (batches resolve in order: the accumulator is summed in sample order) --
// the path-length tail of a batch runs under the bulk of the next ones instead of on a nearly empty GPU.
// wf_flush() does the same for whatever is left at crt_sync; every call that reads or changes state flushes.
struct WfRun {
    bool live = false;              // pipes are forked; every pipe has one outstanding chunk in status buffer [cur]
    int K = 0;
    uint32_t P = 0, Pp = 0, list_cap = 0, trace_blocks = 0;
    uint32_t seg_wps[kWfRing] = {};             // per batch id: work items per shard / in total
    unsigned long long seg_total[kWfRing] = {};
    WfPipe pipes[crt_ctx::kMaxPipes];
    std::vector<WfBatch> open;                  // unresolved batches, oldest first; back() = the newest
    bool queue_left[kWfRing] = {};              // pipe 0's view: that batch's queue still holds work
    bool work_left = false;                     // any of them
    unsigned long long left_its = ~0ull;        // iterations until the newest batch's queue is dry (estimate)
    uint32_t rate_it = 0;                       // pipe 0's iteration and the newest batch's work consumed at its last status
    unsigned long long rate_consumed = 0;
    uint32_t listed_until[kWfRing][crt_ctx::kMaxPipes] = {};   // per id and pipe: launches of iterations < this may look at that queue
};

struct WfConfig {
    uint32_t tiles_x, tiles_y, npix_padded, P, Pp, list_cap, work_per_shard;
    int K;
    unsigned long long work_total;
    size_t npix, list_per_pipe;
};

WfConfig wf_config(crt_ctx *c, uint32_t n)
{
    WfConfig g{};
    g.tiles_x = (c->tw + 7) / 8; g.tiles_y = (c->th + 7) / 8;
    g.npix_padded = g.tiles_x * g.tiles_y * 64u;
    g.npix = (size_t)c->tw * c->th;
    g.work_total = (unsigned long long)n * g.npix_padded;
    // pool: about 1/4 of the batch's paths in flight, between 1 M and 8 M slots (measured best on S2 with up to
    // four batches in flight: whole frame and 1/2, 1/4, 1/8 shares, profiles/r01_steady_pool.log)
    uint32_t P = c->wf_pool ? c->wf_pool : (uint32_t)std::min<unsigned long long>(1u << 23, std::max<unsigned long long>(1u << 20, g.work_total / 4u));
    // (an explicit pool may exceed one batch's work: several batches share it, but never more than the ring holds)
    const unsigned long long most = c->wf_pool ? g.work_total * (unsigned long long)std::max(1, c->wf_ring - 1) : g.work_total;
    if ((unsigned long long)P > most) P = (uint32_t)most;
    // Two (or more) half-pools on separate streams: one half's shade pass (an HBM stream) overlaps
    // the other half's traversal (latency-bound), measured +8 % on S2.  Small jobs keep one pipe.
    int K = std::max(1, std::min(c->wf_pipes, (int)crt_ctx::kMaxPipes));
    if (P < (1u << 18) || (c->wf_pool == 0 && g.work_total < 6000000ull)) K = 1;   // (1-2 spp of a 1080p frame: 7-10 % faster on one pipe)
    // slots per pipe: a whole number of shade blocks for every one of the 64 shards when possible
    // (measured: a 1/8 strip takes 13.0 ms with such a pool and 15.1 ms with one 0.4 % smaller)
    uint32_t Pp = P / (uint32_t)K;
    Pp = Pp >= 16384u ? (Pp / 16384u) * 16384u : ((Pp + 255u) & ~255u);
    g.K = K; g.Pp = Pp; g.P = Pp * (uint32_t)K;
    // list capacity per shard: any shade block size >= 64 maps at most ceil(blocks/shards) blocks to a shard
    g.list_cap = ((Pp / 64u + kWfShards - 1) / kWfShards) * 64u + 256u;
    g.work_per_shard = (uint32_t)((g.work_total + kWfShards - 1) / kWfShards);
    g.work_per_shard = (g.work_per_shard + 63u) & ~63u;
    g.list_per_pipe = (size_t)8 * g.list_cap * kWfShards;                 // [2 parities][4 classes]
    return g;
}

int wf_resolve_batch(crt_ctx *c, const WfBatch &b)
{
    WfRun &r = *c->run;
    WfParams R = r.pipes[0].W;
    R.batch_id = b.id; R.n_samples = b.n;
    HIPCHK(c, wf_launch_resolve(R, b.last_sample, c->stream));
    c->last_launches++;
    return CRT_OK;
}

// Iterations are enqueued in chunks; after each chunk the small control blocks are copied back
// (asynchronously) so the host can tell how far the pool has drained.  One chunk is always
// enqueued AHEAD of the status being waited for, so the GPU never idles on the host; the
// price is at most one chunk of nearly empty iterations at the end.
int wf_enqueue_chunk(crt_ctx *c, int p, int buf)
{
    WfPipe &pp = c->run->pipes[p];
    pp.W.tail_bound = pp.tail_bound;
    for (uint32_t k = 0; k < pp.chunk; k++, pp.it++) {
        pp.W.evict_mask = pp.evict_next;
        HIPCHK(c, wf_launch_shade(pp.W, pp.it, pp.stream));
        if (pp.evict_next) {                                     // k_wf_finish may start once this launch is through
            HIPCHK(c, hipEventRecord(c->ev_evict[p], pp.stream));
            pp.evict_next = 0; pp.W.evict_mask = 0;
        }
        if (c->time_kernels) {
            size_t need = 2 * (size_t)(c->last_trace_kernel_launches + 1);
            while (c->kev.size() < need) {
                hipEvent_t e;
                HIPCHK(c, hipEventCreate(&e));
                c->kev.push_back(e);
            }
            HIPCHK(c, hipEventRecord(c->kev[need - 2], pp.stream));
        }
        HIPCHK(c, wf_launch_trace(pp.W, pp.it, pp.blocks_now, pp.stream));
        if (c->time_kernels) {
            HIPCHK(c, hipEventRecord(c->kev[2 * (size_t)c->last_trace_kernel_launches + 1], pp.stream));
            c->last_trace_kernel_launches++;
        }
        c->last_launches += 2;
        c->last_iterations++;
    }
    pp.it_end[buf] = pp.it;
    for (uint32_t k = 0; k < pp.W.seg_n; k++) c->run->listed_until[pp.W.seg_order[k]][p] = pp.it;
    HIPCHK(c, hipMemcpyAsync(c->h_ctl[p][buf], pp.W.ctl, sizeof(WfCtl), hipMemcpyDeviceToHost, pp.stream));
    HIPCHK(c, hipMemcpyAsync(c->h_wq[p][buf], c->w_wq.p, kWfRing * sizeof(WfWorkQ), hipMemcpyDeviceToHost, pp.stream));
    HIPCHK(c, hipEventRecord(c->ev_ctl[p][buf], pp.stream));
    return CRT_OK;
}

// After every pipe's evicting shade launch, run the evicted paths of batch `b` to their end and resolve the
// batch -- on the context's stream, which has nothing else to do while the pipes work (a stream of its own
// would be a fifth one, and streams beyond the hardware queues share one: a pipe queued behind a wait stalls).
int wf_finish_side(crt_ctx *c, const WfBatch &b, uint32_t max_paths, uint32_t paths_per_wave)
{
    WfRun &r = *c->run;
    for (int p = 0; p < r.K; p++) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_evict[p], 0));
    for (int p = 0; p < r.K; p++) {
        WfParams F = r.pipes[p].W;
        F.batch_id = b.id;
        F.tail_bound = paths_per_wave;
        HIPCHK(c, wf_launch_finish(F, max_paths, c->stream));
        c->last_launches++;
    }
    return wf_resolve_batch(c, b);
}

// The oldest batch is resolved: the next one becomes the one whose survivors the shade launches count.
void wf_pop_oldest(crt_ctx *c)
{
    WfRun &r = *c->run;
    r.open.erase(r.open.begin());
    const uint32_t oldest = r.open.empty() ? 0u : r.open.front().id;
    for (int p = 0; p < r.K; p++) {
        WfPipe &pp = r.pipes[p];
        pp.W.oldest_id = oldest;
        pp.old_from = pp.it; pp.old_valid = false; pp.old = 0;
    }
}

// Which queues the next chunks re-arm from: the open batches whose queue still holds work, oldest first.
void wf_set_queues(crt_ctx *c)
{
    WfRun &r = *c->run;
    uint32_t order[kWfRing], n = 0;
    for (const WfBatch &b : r.open) if (r.queue_left[b.id] && n < kWfRing) order[n++] = b.id;
    if (n == 0) order[n++] = r.open.empty() ? 0u : r.open.back().id;       // (all dry: any valid entry)
    for (int p = 0; p < r.K; p++) {
        for (uint32_t k = 0; k < kWfRing; k++) r.pipes[p].W.seg_order[k] = order[k < n ? k : n - 1];
        r.pipes[p].W.seg_n = n;
    }
}

// Enqueue iterations and read statuses until the newest batch is parked (to_end = false: its queue is nearly
// empty and there is room for another batch) or everything is finished and resolved (to_end = true).
int wf_drive(crt_ctx *c, bool to_end)
{
    WfRun &r = *c->run;
    const int K = r.K;
    const unsigned long long evict_at = std::min<unsigned long long>(c->wf_finish_at, kWfSideCap);
    const unsigned long long flush_at = std::min<unsigned long long>(c->wf_flush_at, kWfSideCap);
    int active = 0;
    for (int p = 0; p < K; p++) { r.pipes[p].done = false; active++; }
    bool oldest_evicting = false, all_evicting = false;
    while (active > 0) {
        for (int p = 0; p < K; p++)
            if (!r.pipes[p].done) { int rc = wf_enqueue_chunk(c, p, r.pipes[p].cur ^ 1); if (rc) return rc; }   // speculative
        // the chunk just enqueued carried the eviction of the oldest batch's last paths: finish and resolve it
        if (oldest_evicting) {
            int rc = wf_finish_side(c, r.open.front(), kWfSideCap, c->wf_side_ppw);
            if (rc) return rc;
            wf_pop_oldest(c);
            oldest_evicting = false;
        }
        for (int p = 0; p < K; p++) {
            WfPipe &pp = r.pipes[p];
            if (pp.done) continue;
            HIPCHK(c, hipEventSynchronize(c->ev_ctl[p][pp.cur]));
            const uint32_t it_seen = pp.it_end[pp.cur];          // the status covers iterations < it_seen
            pp.it_confirmed = it_seen;
            const WfBatch &newest = r.open.back();
            if (it_seen <= newest.from_it[p]) { pp.cur ^= 1; continue; }          // from before the newest batch began
            auto seg_left = [&](uint32_t id, unsigned long long &consumed) {
                bool left = false;
                consumed = 0;
                for (uint32_t sidx = 0; sidx < kWfShards; sidx++) {
                    const unsigned long long lo = (unsigned long long)sidx * r.seg_wps[id];
                    const unsigned long long size = lo < r.seg_total[id] ? std::min<unsigned long long>(r.seg_wps[id], r.seg_total[id] - lo) : 0;
                    const unsigned long long cur = c->h_wq[p][pp.cur][id].work[sidx].cur;
                    if (cur < size) left = true;
                    consumed += std::min(cur, size);
                }
                return left;
            };
            unsigned long long newest_consumed = 0;
            for (const WfBatch &b : r.open) {
                unsigned long long cons = 0;
                const bool left = seg_left(b.id, cons);
                if (!left) pp.dry[b.id] = true;
                if (&b == &newest) newest_consumed = cons;
                if (p == 0) r.queue_left[b.id] = left;          // monotone within a batch: once false it stays false
            }
            if (p == 0) {
                r.work_left = false;
                for (const WfBatch &b : r.open) r.work_left = r.work_left || r.queue_left[b.id];
                // How many iterations until the newest batch's queue is dry.  Chunks shrink as that comes close when
                // no further batch can take over: what is enqueued ahead of the status that shows the queues empty
                // runs on a nearly empty pool, and the host needs only ~20 us per launch to keep up.
                const uint32_t its = it_seen - r.rate_it;
                if (!r.queue_left[newest.id]) r.left_its = 0;
                else if (its > 0 && newest_consumed > r.rate_consumed) {
                    const unsigned long long per_it = (newest_consumed - r.rate_consumed) / its;
                    r.left_its = (r.seg_total[newest.id] - newest_consumed) / std::max<unsigned long long>(per_it, 1);
                }
                if (to_end || r.open.size() >= (size_t)c->wf_ring) {
                    const uint32_t chunk = r.left_its >= 6 ? (uint32_t)c->wf_chunk : 1u;
                    for (int q = 0; q < K; q++) r.pipes[q].chunk = chunk;
                }
                r.rate_it = it_seen; r.rate_consumed = newest_consumed;
                wf_set_queues(c);
            }
            const WfCtl *hc = c->h_ctl[p][pp.cur];
            if (hc->dropped) return fail(c, CRT_EDEVICE, "wavefront pipeline: a capacity guard dropped %u paths (pipe %d)", hc->dropped, p);
            unsigned long long rays = 0, old = 0;
            uint32_t bound = 0;
            for (uint32_t sidx = 0; sidx < kWfShards; sidx++) {
                const WfShard &sh = hc->shard[(it_seen - 1) & 3u][sidx];
                // per-shard bound for later iterations: slots never change shard and none are re-armed once the
                // queues are empty, so no list of a shard can ever grow beyond the slots alive in it now
                uint32_t alive_here = 0;
                for (int k = 0; k < 4; k++) { rays += sh.n[k]; alive_here += sh.n[k]; }
                bound = std::max(bound, alive_here);
                old += sh.old;
            }
            pp.rays = rays; pp.any = true;
            if (it_seen > pp.old_from) { pp.old = old; pp.old_valid = true; }
            if (getenv("CRT_DEBUG")) fprintf(stderr, "[crt] pipe %d it %u rays %llu old %llu open %zu work_left %d bound %u\n", p, it_seen, rays, old, r.open.size(), (int)r.work_left, bound);
            // pipe 0's view of the queues can lag the others by a chunk; a pipe with no rays while work
            // may be left simply keeps going (its dead slots re-arm as soon as they see work)
            if (!r.work_left && rays == 0) { pp.done = true; pp.old = 0; pp.old_valid = true; pp.cur ^= 1; active--; continue; }   // every alive slot lists a ray
            if (!to_end) {
                // Park: the newest batch's queue is nearly dry -- NEARLY, so that the next call can publish its
                // queue while this one still holds work and the chunk already enqueued ahead keeps the GPU busy --
                // and the ring has room for another batch.
                if (r.open.size() < (size_t)c->wf_ring && r.left_its < (unsigned long long)c->wf_park_its) { pp.done = true; pp.cur ^= 1; active--; continue; }
            } else if (!r.work_left && !all_evicting) {
                if (rays < std::min<unsigned long long>((unsigned long long)r.Pp / 4u, 65536ull) && c->wf_tail_walk) {
                    // The tail: no path can start any more, so ray counts only shrink from here.  Shade walks
                    // the ray lists instead of the whole pool and the grids shrink.
                    pp.tail_bound = std::max<uint32_t>(64u, (bound + 63u) & ~63u);
                    pp.blocks_now = (uint32_t)std::min<unsigned long long>(r.trace_blocks, std::max<unsigned long long>(64, rays / 32u + 64u));
                }
            }
            if (pp.it - newest.from_it[p] > 100000u) return fail(c, CRT_EDEVICE, "wavefront pipeline did not drain");
            pp.cur ^= 1;
        }
        // The oldest batch (when it is not the newest): its queue is dry for every pipe and few of its paths are
        // left (alive slots only shrink, so they still fit when the launch runs) -- the next shade launch of every
        // pipe moves them to the side pool; or none are left at all -- it is resolved right away.
        if (r.open.size() > 1 && !oldest_evicting && !all_evicting && active > 0) {
            const WfBatch &ob = r.open.front();
            bool ready = true;
            unsigned long long old = 0, old_max = 0;
            for (int p = 0; p < K; p++) {
                const WfPipe &pp = r.pipes[p];
                if (pp.done) continue;                           // (a drained pipe holds no path at all)
                ready = ready && pp.old_valid && pp.dry[ob.id];
                old += pp.old; old_max = std::max(old_max, pp.old);
            }
            if (ready && old == 0) { int rc = wf_resolve_batch(c, ob); if (rc) return rc; wf_pop_oldest(c); }
            else if (ready && old_max <= kWfSideCap && old <= evict_at * (unsigned)K) {
                for (int p = 0; p < K; p++) if (!r.pipes[p].done) r.pipes[p].evict_next = 1u << ob.id;
                oldest_evicting = true;
            }
        }
        // to_end: once few paths are left altogether, everything alive goes to the side pools and the pool is done
        if (to_end && !r.work_left && !all_evicting && !oldest_evicting && active > 0 && flush_at > 0) {
            bool ready = true;
            for (int p = 0; p < K; p++) {
                if (r.pipes[p].done) continue;
                ready = ready && r.pipes[p].any && r.pipes[p].rays <= flush_at;
            }
            if (ready) {
                uint32_t mask = 0;
                for (const WfBatch &b : r.open) mask |= 1u << b.id;
                for (int p = 0; p < K; p++) {
                    WfPipe &pp = r.pipes[p];
                    if (pp.done) continue;
                    pp.evict_next = mask;
                    const uint32_t chunk = pp.chunk;
                    pp.chunk = 1;                                // one more iteration: its shade launch empties the pool
                    int rc = wf_enqueue_chunk(c, p, pp.cur ^ 1);
                    pp.chunk = chunk;
                    if (rc) return rc;
                    pp.done = true; pp.rays = 0; pp.cur ^= 1;
                }
                active = 0;
                all_evicting = true;
            }
        }
    }
    bool empty = !r.work_left;
    for (int p = 0; p < K; p++) empty = empty && r.pipes[p].rays == 0;
    if (to_end || empty) {
        if (!all_evicting) {
            // everything enqueued for the pipes comes before the resolve passes on the context's stream
            for (int p = 0; p < K; p++) {
                HIPCHK(c, hipEventRecord(c->ev_join[p], r.pipes[p].stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[p], 0));
            }
        }
        while (!r.open.empty()) {                                // in order: the accumulator is summed in sample order
            // (when everything was evicted nothing else is running: few paths per wave end sooner)
            int rc = all_evicting ? wf_finish_side(c, r.open.front(), (uint32_t)flush_at, c->wf_flush_ppw) : wf_resolve_batch(c, r.open.front());
            if (rc) return rc;
            wf_pop_oldest(c);
        }
        if (c->counting) {
            // fold the pipes' counters into the context's
            HIPCHK(c, hipStreamSynchronize(c->stream));
            unsigned long long tot[CRT_NCOUNTERS];
            HIPCHK(c, hipMemcpy(tot, c->d_counters.p, sizeof tot, hipMemcpyDeviceToHost));
            for (int p = 0; p < K; p++) {
                HIPCHK(c, hipMemcpy(c->h_ctl[p][0], c->w_ctl[p].p, sizeof(WfCtl), hipMemcpyDeviceToHost));
                for (int k = 0; k < CRT_NCOUNTERS; k++) tot[k] += c->h_ctl[p][0]->counters[k];
                for (int k = 0; k < 8; k++) c->probes[k] += c->h_ctl[p][0]->counters[8 + k];
                HIPCHK(c, hipMemset(&c->w_ctl[p].p->counters[0], 0, sizeof(unsigned long long) * CRT_NCOUNTERS_DEV));
            }
            HIPCHK(c, hipMemcpy(c->d_counters.p, tot, sizeof tot, hipMemcpyHostToDevice));
        }
    }
    if (to_end || empty) {
        // (checked by wf_check_dropped after the caller's stream synchronisation)
        for (int p = 0; p < K; p++)
            HIPCHK(c, hipMemcpyAsync(&c->h_dropped[p], &c->w_ctl[p].p->dropped, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        // the pool is empty; the next batch sets the pipes up afresh (after everything enqueued here)
        for (int p = 0; p < K; p++) {
            HIPCHK(c, hipEventRecord(c->ev_join[p], r.pipes[p].stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[p], 0));
        }
        r.live = false;
    }
    return CRT_OK;
}

// After a flush and a synchronisation of the context's stream: did a device-side capacity guard drop a path?
int wf_check_dropped(crt_ctx *c)
{
    if (!c->h_dropped) return CRT_OK;
    for (int p = 0; p < crt_ctx::kMaxPipes; p++)
        if (c->h_dropped[p]) {
            const uint32_t n = c->h_dropped[p];
            c->h_dropped[p] = 0;
            return fail(c, CRT_EDEVICE, "wavefront pipeline: a capacity guard dropped %u paths (pipe %d); the frame is incomplete", n, p);
        }
    return CRT_OK;
}

// Finish whatever the pipeline still holds (no-op when nothing is parked).
int wf_flush(crt_ctx *c)
{
    if (!c->run || !c->run->live) return CRT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = wf_drive(c, true);
    if (rc == CRT_OK && c->last_timed) HIPCHK(c, hipEventRecord(c->ev1, c->stream));   // crt_last_trace_ms covers the stragglers too
    return rc;
}

// One batch of n samples through the wavefront pipeline (asynchronous except for the small
// control-block readbacks that tell how far the pool has drained).
int wf_trace_batch(crt_ctx *c, uint32_t n)
{
    if (!c->run) c->run = new WfRun();
    WfRun &r = *c->run;
    const WfConfig g = wf_config(c, n);
    if (g.npix == 0 || n == 0) { int rc = wf_flush(c); c->sample += n; return rc; }
    // counting folds counters on the host after every batch; otherwise batches end parked
    const bool defer = c->wf_defer && !c->counting;
    const size_t staging_elems = (size_t)n * g.npix;
    const uint32_t side_slots = kWfRing * (uint32_t)crt_ctx::kMaxPipes * kWfSideCap;   // side pools first, then the pool
    if (r.live && (g.K != r.K || g.Pp != r.Pp || c->w_staging[0].n < staging_elems || r.open.size() >= (size_t)c->wf_ring)) {
        int rc = wf_flush(c);
        if (rc) return rc;
    }
    int rc = wf_ensure(c, (size_t)g.P + side_slots, staging_elems, g.list_per_pipe * (size_t)g.K);
    if (rc) return rc;
    r.rate_consumed = 0;
    r.left_its = ~0ull;
    if (!r.live) {
        r.K = g.K; r.P = g.P; r.Pp = g.Pp; r.list_cap = g.list_cap;
        r.trace_blocks = (uint32_t)c->num_cu * c->wf_waves_per_cu;
        r.open.clear();
        WfBatch nb;
        nb.n = n; nb.last_sample = c->sample + n; nb.id = 0;
        r.open.push_back(nb);
        r.rate_it = 0;
        r.seg_total[0] = g.work_total; r.seg_wps[0] = g.work_per_shard;
        for (uint32_t b = 0; b < kWfRing; b++) { r.queue_left[b] = false; for (int p = 0; p < crt_ctx::kMaxPipes; p++) r.listed_until[b][p] = 0; }
        r.queue_left[0] = r.work_left = true;
        for (int p = 0; p < r.K; p++) {
            r.pipes[p] = WfPipe();
            r.pipes[p].chunk = (uint32_t)c->wf_chunk;
            WfParams &W = r.pipes[p].W;
            W.sc = c->sc;
            W.ray_o = c->w_ray_o.p; W.ray_d = c->w_ray_d.p; W.sh_d = c->w_sh_d.p; W.beta = c->w_beta.p;
            W.radiance = c->w_radiance.p; W.nee = c->w_nee.p; W.rng = c->w_rng.p; W.misc = c->w_misc.p;
            W.hit = c->w_hit.p; W.vis = c->w_vis.p;
            for (int b = 0; b < 2; b++)
                for (int k = 0; k < 4; k++)
                    W.list[b][k] = c->w_list_ext.p + g.list_per_pipe * (size_t)p + (size_t)(b * 4 + k) * g.list_cap * kWfShards;
            for (uint32_t b = 0; b < kWfRing; b++) {
                W.staging[b] = c->w_staging[b].p;
                W.side_base[b] = (b * (uint32_t)crt_ctx::kMaxPipes + (uint32_t)p) * kWfSideCap;
                W.seg[b] = WfSeg{0, 64, 0};
                W.seg_order[b] = 0;
            }
            W.batch_id = 0; W.oldest_id = 0; W.keep_pool = 0; W.evict_mask = 0;
            W.ctl = c->w_ctl[p].p; W.wq = c->w_wq.p;
            W.slot_base = side_slots + g.Pp * (uint32_t)p; W.reset_wq = (p == 0) ? 1u : 0u;
            W.P = g.Pp; W.x0 = c->x0; W.y0 = c->y0; W.tw = c->tw; W.th = c->th;
            W.band = c->band; W.stride = c->stride; W.phase = c->phase;
            W.tiles_x = g.tiles_x; W.tiles_y = g.tiles_y; W.npix_padded = g.npix_padded;
            W.list_cap = g.list_cap;
            W.seg[0] = WfSeg{g.work_total, g.work_per_shard, c->sample + 1};
            W.seg_n = 1;
            W.n_samples = n;
            W.accum = accum_ptr(c); W.rgba = rgba_ptr(c);
            W.tea = c->w_tea.p;
            W.count = c->counting ? 1u : 0u;
            W.overflow_lanes = (uint32_t)c->num_cu * c->wf_waves_per_cu * 64u;
            W.stack_overflow = c->w_overflow.p + (size_t)p * W.overflow_lanes * 64u;
            if (!c->pipe_stream[p]) {
                // Streams beyond the hardware queues (4 by default) share one, and two pipes sharing a queue do not
                // overlap at all (measured: 95 instead of 77 ms per S2 frame when the caller's framework had taken
                // the queues first).  The runtime keeps separate queues per priority level and frameworks create
                // their stream pools at the default level, so the pipes take the high one -- all of them the same,
                // an uneven pair measured 4-9 % slower.
                int least = 0, greatest = 0;
                HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIPCHK(c, hipStreamCreateWithPriority(&c->pipe_stream[p], hipStreamNonBlocking, greatest));
            }
            r.pipes[p].stream = c->pipe_stream[p];
            r.pipes[p].blocks_now = r.trace_blocks;
        }
        // The context's stream is the control stream: it resets the work queue and forks the pipes (and,
        // later, finishes stragglers and resolves).  The pipes run on their own streams.
        HIPCHK(c, wf_launch_init(r.pipes[0].W, c->stream));
        HIPCHK(c, wf_launch_tea(r.pipes[0].W, c->w_tea.p, c->stream));          // per-pixel RNG seed words of this tile
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        for (int p = 0; p < r.K; p++) {
            HIPCHK(c, hipStreamWaitEvent(r.pipes[p].stream, c->ev_fork, 0));
            if (p > 0) HIPCHK(c, wf_launch_init(r.pipes[p].W, r.pipes[p].stream));
        }
        // the first chunk of every pipe (from here on one chunk per pipe is always outstanding)
        for (int p = 0; p < r.K; p++) { rc = wf_enqueue_chunk(c, p, 0); if (rc) return rc; }
        r.live = true;
    } else {
        // The parked batches keep their slots, queues and staging buffers; this batch takes the next id and its
        // work flows into the slots that are free once the older queues are dry.
        WfBatch nb;
        nb.n = n; nb.last_sample = c->sample + n; nb.id = (r.open.back().id + 1u) % kWfRing;
        const uint32_t id = nb.id;
        r.seg_total[id] = g.work_total; r.seg_wps[id] = g.work_per_shard;
        r.queue_left[id] = r.work_left = true;
        for (int p = 0; p < r.K; p++) {
            WfPipe &pp = r.pipes[p];
            nb.from_it[p] = pp.it;
            pp.W.seg[id] = WfSeg{g.work_total, g.work_per_shard, c->sample + 1};
            pp.W.n_samples = n;
            pp.W.batch_id = id; pp.W.keep_pool = 1;
            pp.tail_bound = 0; pp.blocks_now = r.trace_blocks;
            pp.any = false; pp.evict_next = 0; pp.chunk = (uint32_t)c->wf_chunk;
            pp.dry[id] = false;
            // (while it was the only batch, the oldest was also the newest and nothing was counted: the survivor
            // counts of the oldest batch start with the launches enqueued from here on)
            if (r.open.size() == 1) { pp.old_from = pp.it; pp.old_valid = false; pp.old = 0; }
        }
        r.open.push_back(nb);
        r.rate_it = r.pipes[0].it;
        wf_set_queues(c);
        // This batch's queue and side counters are reset on the control stream.  The batch that used the id
        // before is resolved; a launch still in flight may have that queue in its list, though (enqueued while
        // it held work), and would take the NEW work with the OLD batch's parameters: the reset waits for such
        // a pipe's outstanding chunk.  Otherwise the pipes only wait for the reset, not for each other.
        for (int p = 0; p < r.K; p++)
            if (r.listed_until[id][p] > r.pipes[p].it_confirmed) {
                HIPCHK(c, hipEventRecord(c->ev_join[p], r.pipes[p].stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[p], 0));
            }
        for (int p = 0; p < r.K; p++) HIPCHK(c, wf_launch_init(r.pipes[p].W, c->stream));   // (one block each)
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        for (int p = 0; p < r.K; p++) HIPCHK(c, hipStreamWaitEvent(r.pipes[p].stream, c->ev_fork, 0));
    }
    c->sample += n;
    return wf_drive(c, !defer);
}

}  // namespace

extern "C" {

int crt_abi_version(void) { return CRT_ABI_VERSION; }

const char *crt_last_error(crt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int crt_create(crt_ctx **out, int device_ordinal)
{
    if (!out) return fail(nullptr, CRT_EINVAL, "crt_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, CRT_EDEVICE, "crt_create: no HIP device (%s); there is no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_ordinal < 0 || device_ordinal >= ndev)
        return fail(nullptr, CRT_EINVAL, "crt_create: device %d out of range (0..%d)", device_ordinal, ndev - 1);
    crt_ctx *c = new crt_ctx();
    c->device = device_ordinal;
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = c->d_counters.alloc(CRT_NCOUNTERS)) != hipSuccess ||
        (e = hipMemset(c->d_counters.p, 0, CRT_NCOUNTERS * sizeof(unsigned long long))) != hipSuccess) {
        int rc = fail(nullptr, CRT_EDEVICE, "crt_create: %s", hipGetErrorString(e));
        crt_destroy(c);
        return rc;
    }
    c->stream = c->own_stream;
    *out = c;
    return CRT_OK;
}

void crt_destroy(crt_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    // parked work is abandoned, but every stream must have drained before the buffers go
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int p = 0; p < crt_ctx::kMaxPipes; p++) if (c->pipe_stream[p]) (void)hipStreamSynchronize(c->pipe_stream[p]);

    delete c->run;
    c->d_prim.release(); c->d_primD.release(); c->d_nodes.release(); c->d_nodes4.release(); c->d_nodes4q.release(); c->d_nodes8q.release(); c->d_lights.release(); c->w_overflow.release();
    c->d_slot_of_index.release(); c->d_spectra.release(); c->d_cie.release();
    c->d_accum.release(); c->d_rgba.release(); c->d_counters.release();
    c->w_ray_o.release(); c->w_ray_d.release(); c->w_sh_d.release(); c->w_beta.release(); c->w_radiance.release();
    c->w_nee.release(); for (uint32_t b = 0; b < kWfRing; b++) c->w_staging[b].release(); c->w_rng.release(); c->w_misc.release(); c->w_hit.release();
    c->w_vis.release(); c->w_list_ext.release(); c->w_tea.release(); c->w_wq.release();
    for (int p = 0; p < crt_ctx::kMaxPipes; p++) {
        c->w_ctl[p].release();
        for (int b = 0; b < 2; b++) {
            if (c->h_ctl[p][b]) (void)hipHostFree(c->h_ctl[p][b]);
            if (c->ev_ctl[p][b]) (void)hipEventDestroy(c->ev_ctl[p][b]);
        }
        if (c->pipe_stream[p]) (void)hipStreamDestroy(c->pipe_stream[p]);
        if (c->ev_join[p]) (void)hipEventDestroy(c->ev_join[p]);
        if (c->ev_evict[p]) (void)hipEventDestroy(c->ev_evict[p]);
    }
    for (int p = 0; p < crt_ctx::kMaxPipes; p++)
        for (int b = 0; b < 2; b++) if (c->h_wq[p][b]) (void)hipHostFree(c->h_wq[p][b]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->h_dropped) (void)hipHostFree(c->h_dropped);

    for (hipEvent_t e : c->kev) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int crt_upload_scene(crt_ctx *c, const void *primitives, size_t nprim, const void *lights, size_t nlight,
                     const float *spectra, size_t nspectra, const float *cie, const float camera[16])
{
    if (!c) return CRT_EINVAL;
    if ((!primitives && nprim) || !lights || !spectra || !cie || !camera)
        return fail(c, CRT_EINVAL, "crt_upload_scene: NULL buffer");
    if (nlight < 1) return fail(c, CRT_EINVAL, "crt_upload_scene: at least one light record is required");
    if (nspectra < 1 || nspectra > 0x3FFF) return fail(c, CRT_EINVAL, "crt_upload_scene: nspectra must be 1..16383");
    if (nprim >= (1u << 28)) return fail(c, CRT_EINVAL, "crt_upload_scene: too many primitives");
    if (!(camera[11] >= 1.0f && camera[12] >= 1.0f && camera[11] <= 65536.0f && camera[12] <= 65536.0f))
        return fail(c, CRT_EINVAL, "crt_upload_scene: camera width/height (floats 11,12) must be 1..65536");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));

    std::vector<HostPrim> prims(nprim), lts(nlight);
    for (size_t i = 0; i < nprim; i++) {
        prims[i] = read_prim((const uint8_t *)primitives, i);
        const HostPrim &p = prims[i];
        if (p.category > 2u) return fail(c, CRT_EINVAL, "primitive %zu: category %u not in {0,1,2}", i, p.category);
        if (p.material > 2u) return fail(c, CRT_EINVAL, "primitive %zu: material %u not in {0,1,2}", i, p.material);
        if (p.index != (uint32_t)i)
            return fail(c, CRT_EINVAL, "primitive %zu: data4.w (index) is %u, must equal the array position "
                                       "(src/main.js:124,133)", i, p.index);
        if (p.emission >= nspectra || p.reflectance >= nspectra)
            return fail(c, CRT_EINVAL, "primitive %zu: spectrum index out of range", i);
    }
    for (size_t i = 0; i < nlight; i++) {
        lts[i] = read_prim((const uint8_t *)lights, i);
        if (lts[i].emission >= nspectra) return fail(c, CRT_EINVAL, "light %zu: emission index out of range", i);
    }
    // from here on the old scene is gone: a failure below must not leave a context that can still trace
    c->have_scene = false;
    c->accel_mode = -1;
    c->prims.swap(prims);
    c->lights.swap(lts);
    std::memcpy(c->camera, camera, sizeof c->camera);
    c->W = (uint32_t)camera[11];                                 // ComputeShader.wgsl:85
    c->H = (uint32_t)camera[12];

    DevScene &S = c->sc;
    S = DevScene{};
    S.W = c->W; S.H = c->H;
    S.nspectra = (uint32_t)nspectra;
    S.nlight = (uint32_t)nlight;
    S.inv_nlight = 1.0f / (float)S.nlight;                       // :372-373
    S.hit_pad = scene_hit_pad(c->prims, c->camera);
    camera_frame(c->camera, S.cam);

    HIPCHK(c, c->d_spectra.alloc(nspectra * kNLambda));
    HIPCHK(c, hipMemcpy(c->d_spectra.p, spectra, nspectra * kNLambda * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, c->d_cie.alloc(3 * kNCie));
    HIPCHK(c, hipMemcpy(c->d_cie.p, cie, 3 * kNCie * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float4> hl(nlight * 3);
    for (size_t i = 0; i < nlight; i++) {
        const HostPrim &l = c->lights[i];
        float light_area = length(l.d2) * length(l.d3);          // :363
        hl[3 * i + 0] = float4{l.d1.x, l.d1.y, l.d1.z, bits_f(l.emission)};
        hl[3 * i + 1] = float4{l.d2.x, l.d2.y, l.d2.z, bits_f(l.index)};
        hl[3 * i + 2] = float4{l.d3.x, l.d3.y, l.d3.z, 1.0f / light_area};   // :364
    }
    HIPCHK(c, c->d_lights.alloc(hl.size()));
    HIPCHK(c, hipMemcpy(c->d_lights.p, hl.data(), hl.size() * sizeof(float4), hipMemcpyHostToDevice));
    S.spectra = c->d_spectra.p; S.cie = c->d_cie.p; S.lights = c->d_lights.p;

    c->x0 = 0; c->y0 = 0; c->tw = c->W; c->th = c->H;
    c->band = 0x40000000u; c->stride = 1; c->phase = 0;
    c->accum_bound = nullptr; c->rgba_bound = nullptr;
    c->have_scene = true;                                        // (alloc_tile / zero_state below need it for the error paths of others)
    int rc = alloc_tile(c);
    if (rc == CRT_OK) rc = zero_state(c);
    if (rc) c->have_scene = false;
    return rc;
}

int crt_set_tile(crt_ctx *c, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    if (!c) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_set_tile: upload a scene first");
    if (x0 > x1 || y0 > y1 || x1 > c->W || y1 > c->H)
        return fail(c, CRT_EINVAL, "crt_set_tile: rectangle [%u,%u)x[%u,%u) outside %ux%u", x0, x1, y0, y1, c->W, c->H);
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->x0 = x0; c->y0 = y0; c->tw = x1 - x0; c->th = y1 - y0;
    c->band = 0x40000000u; c->stride = 1; c->phase = 0;
    c->accum_bound = nullptr; c->rgba_bound = nullptr;
    int rc = alloc_tile(c);
    if (rc) return rc;
    return zero_state(c);
}

int crt_set_row_bands(crt_ctx *c, uint32_t band_rows, uint32_t parts, uint32_t part)
{
    if (!c) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_set_row_bands: upload a scene first");
    if (band_rows == 0 || parts == 0 || part >= parts || band_rows > 65536u)
        return fail(c, CRT_EINVAL, "crt_set_row_bands: need band_rows >= 1 and part < parts");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint32_t rows = 0;                                   // rows y of the frame with (y / band) % parts == part
    for (uint32_t b = part; (unsigned long long)b * band_rows < c->H; b += parts)
        rows += std::min<uint32_t>(band_rows, c->H - b * band_rows);
    c->x0 = 0; c->y0 = 0; c->tw = c->W; c->th = rows;
    c->band = band_rows; c->stride = parts; c->phase = part;
    c->accum_bound = nullptr; c->rgba_bound = nullptr;
    int rc = alloc_tile(c);
    if (rc) return rc;
    return zero_state(c);
}

int crt_build_accel(crt_ctx *c, int mode)
{
    if (!c) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_build_accel: upload a scene first");
    if (mode != CRT_ACCEL_NONE && mode != CRT_ACCEL_BVH2 && mode != CRT_ACCEL_LBVH) return fail(c, CRT_EINVAL, "crt_build_accel: unknown mode %d", mode);
    c->want_lbvh = mode == CRT_ACCEL_LBVH;
    if (mode == CRT_ACCEL_LBVH) mode = CRT_ACCEL_BVH2;          // same structure, same kernels
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return upload_geometry(c, mode);
}

int crt_reset(crt_ctx *c)
{
    if (!c) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_reset: upload a scene first");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    return zero_state(c);
}

int crt_trace(crt_ctx *c, uint32_t n_samples)
{
    if (!c) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_trace: upload a scene first");
    if (c->accel_mode < 0) return fail(c, CRT_ESTATE, "crt_trace: call crt_build_accel first");
    HIPCHK(c, hipSetDevice(c->device));
    if ((size_t)c->tw * c->th != 0 && (!accum_ptr(c) || !rgba_ptr(c)))
        return fail(c, CRT_ENOMEM, "crt_trace: the tile's output buffers are not allocated (an earlier crt_set_tile / crt_set_row_bands failed)");
    TraceParams P{};
    P.sc = c->sc;
    P.x0 = c->x0; P.y0 = c->y0; P.tw = c->tw; P.th = c->th;
    P.band = c->band; P.stride = c->stride; P.phase = c->phase;
    P.accum = accum_ptr(c); P.rgba = rgba_ptr(c);
    P.counters = c->counting ? c->d_counters.p : nullptr;
    P.tiles_x = (c->tw + 7) / 8; P.tiles_y = (c->th + 7) / 8;        // main.js:606-610
    c->last_launches = 0;
    c->last_timed = true;
    c->last_iterations = 0;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    uint32_t left = n_samples;
    if (c->pipeline == 1 && c->accel_mode == CRT_ACCEL_BVH2) {
        // batch so that the staging buffer stays below ~6 GB and work ids fit 32 bits
        const size_t npix = std::max<size_t>((size_t)c->tw * c->th, 1);
        uint32_t cap = (uint32_t)std::max<size_t>(1, std::min<size_t>(256, (size_t)6e9 / (npix * 16)));
        if (c->spp_per_launch) cap = std::min(cap, c->spp_per_launch);
        while (left) {
            uint32_t n = std::min(left, cap);
            int rc = wf_trace_batch(c, n);
            if (rc) return rc;
            left -= n;
        }
    } else {
        { int rc_ = wf_flush(c); if (rc_) return rc_; }
        uint32_t chunk = c->spp_per_launch ? c->spp_per_launch : 8u;
        while (left) {
            uint32_t n = std::min(left, chunk);
            P.first_sample = c->sample + 1;                               // UpdateVariables.wgsl: sample++ first
            P.n_samples = n;
            HIPCHK(c, launch_trace(P, c->counting, c->accel_mode == CRT_ACCEL_NONE, c->stream));
            c->sample += n;
            left -= n;
            c->last_launches++;
        }
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    return CRT_OK;
}

int crt_sync(crt_ctx *c)
{
    if (!c) return CRT_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return wf_check_dropped(c);
}

int crt_sample_count(crt_ctx *c, uint32_t *out)
{
    if (!c || !out) return CRT_EINVAL;
    *out = c->sample;
    return CRT_OK;
}

int crt_tile(crt_ctx *c, uint32_t out[4])
{
    if (!c || !out) return CRT_EINVAL;
    out[0] = c->x0; out[1] = c->y0; out[2] = c->tw; out[3] = c->th;
    return CRT_OK;
}

int crt_read_accum(crt_ctx *c, float *out)
{
    if (!c || !out) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_read_accum: no scene");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    size_t n = (size_t)c->tw * c->th;
    if (n) HIPCHK(c, hipMemcpyAsync(out, accum_ptr(c), n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return wf_check_dropped(c);
}

int crt_read_rgba8(crt_ctx *c, uint8_t *out)
{
    if (!c || !out) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_read_rgba8: no scene");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    size_t n = (size_t)c->tw * c->th;
    if (n) HIPCHK(c, hipMemcpyAsync(out, rgba_ptr(c), n * sizeof(uchar4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return wf_check_dropped(c);
}

int crt_write_accum(crt_ctx *c, const float *in, uint32_t sample)
{
    if (!c || !in) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_write_accum: no scene");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    size_t n = (size_t)c->tw * c->th;
    if (n) HIPCHK(c, hipMemcpyAsync(accum_ptr(c), in, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->sample = sample;
    return CRT_OK;
}

int crt_device_buffers(crt_ctx *c, void **accum_dev, void **rgba8_dev)
{
    if (!c) return CRT_EINVAL;
    if (accum_dev) *accum_dev = accum_ptr(c);
    if (rgba8_dev) *rgba8_dev = rgba_ptr(c);
    return CRT_OK;
}

int crt_bind_output(crt_ctx *c, void *accum_dev, void *rgba8_dev)
{
    if (!c) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_bind_output: upload a scene first");
    if (((uintptr_t)accum_dev & 15u) || ((uintptr_t)rgba8_dev & 3u))
        return fail(c, CRT_EINVAL, "crt_bind_output: accum must be 16-byte and rgba8 4-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->accum_bound = (float4 *)accum_dev;
    c->rgba_bound = (uchar4 *)rgba8_dev;
    return CRT_OK;
}

int crt_set_stream(crt_ctx *c, void *hip_stream)
{
    if (!c) return CRT_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return CRT_OK;
}

int crt_enable_counters(crt_ctx *c, int on)
{
    if (!c) return CRT_EINVAL;
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    c->counting = on != 0;
    return CRT_OK;
}

int crt_debug_probes(crt_ctx *c, uint64_t out[8])
{
    if (!c || !out) return CRT_EINVAL;
    for (int k = 0; k < 8; k++) out[k] = c->probes[k];
    return CRT_OK;
}

int crt_reset_counters(crt_ctx *c)
{
    if (!c) return CRT_EINVAL;
    for (int k = 0; k < 8; k++) c->probes[k] = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_counters.p, 0, CRT_NCOUNTERS * sizeof(unsigned long long), c->stream));
    return CRT_OK;
}

int crt_counters(crt_ctx *c, uint64_t out[CRT_NCOUNTERS])
{
    if (!c || !out) return CRT_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipMemcpyAsync(out, c->d_counters.p, CRT_NCOUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

int crt_last_trace_ms(crt_ctx *c, float *ms, uint32_t *launches)
{
    if (!c) return CRT_EINVAL;
    if (!c->last_timed) return fail(c, CRT_ESTATE, "crt_last_trace_ms: no crt_trace yet");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float t = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&t, c->ev0, c->ev1));
    c->last_ms = t;
    if (ms) *ms = t;
    if (launches) *launches = c->last_launches;
    return CRT_OK;
}

int crt_last_kernel_ms(crt_ctx *c, float *ms, uint32_t *launches)
{
    if (!c) return CRT_EINVAL;
    if (!c->last_timed) return fail(c, CRT_ESTATE, "crt_last_kernel_ms: no crt_trace yet");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float total = 0.0f;
    uint32_t n = 0;
    if (c->pipeline == 1 && c->accel_mode == CRT_ACCEL_BVH2) {
        if (!c->time_kernels) return fail(c, CRT_ESTATE, "crt_last_kernel_ms: set option time_kernels=1 before crt_trace");
        HIPCHK(c, hipStreamSynchronize(c->stream));
        n = c->last_trace_kernel_launches;
        for (uint32_t i = 0; i < n; i++) {
            float t = 0.0f;
            HIPCHK(c, hipEventElapsedTime(&t, c->kev[2 * (size_t)i], c->kev[2 * (size_t)i + 1]));
            total += t;
        }
        c->last_trace_kernel_launches = 0;               // the next query starts a new interval
    } else {
        HIPCHK(c, hipEventElapsedTime(&total, c->ev0, c->ev1));
        n = c->last_launches;
    }
    if (ms) *ms = total;
    if (launches) *launches = n;
    return CRT_OK;
}

int crt_accel_stats(crt_ctx *c, uint64_t out[8])
{
    if (!c || !out) return CRT_EINVAL;
    const bool wide = c->pipeline == 1 && c->accel_mode == CRT_ACCEL_BVH2 && c->bvh4.n_inner > 0;
    const bool wide8 = wide && c->bvh8q.ok;
    out[4] = wide ? (c->bvh4q.ok ? 16 : 32) : 32;      // bytes of node data per child box tested
    out[5] = wide8 ? 8 : wide ? 4 : 2;                  // node width used by crt_trace
    out[6] = wide8 ? c->bvh8q.n_inner : wide ? c->bvh4.n_inner : c->bvh.n_inner;   // inner nodes of that tree
    out[7] = (uint64_t)c->accel_builder;               // 0: host binned SAH, 1: GPU LBVH
    out[0] = c->bvh.n_inner; out[1] = c->bvh.n_leaves; out[2] = c->bvh.max_depth;
    out[3] = (uint64_t)c->bvh.n_inner * 64u + (uint64_t)c->bvh4.n_inner * 128u + (uint64_t)c->prims.size() * 48u;
    return CRT_OK;
}

int crt_set_option(crt_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return CRT_EINVAL;
    { int rc_ = wf_flush(c); if (rc_) return rc_; }
    if (!std::strcmp(name, "debug_fail_alloc")) { g_fail_alloc_in = value; return CRT_OK; }
    if (!std::strcmp(name, "wf_defer")) { c->wf_defer = value != 0; return CRT_OK; }
    if (!std::strcmp(name, "spp_per_launch")) { c->spp_per_launch = (uint32_t)std::max<int64_t>(0, value); return CRT_OK; }
    if (!std::strcmp(name, "pipeline")) { c->pipeline = value ? 1 : 0; return CRT_OK; }
    if (!std::strcmp(name, "quantize")) { c->quantize = value ? 1 : 0; return CRT_OK; }   // takes effect at crt_build_accel
    if (!std::strcmp(name, "wf_width")) { c->wf_width = value == 4 ? 4 : 8; return CRT_OK; }   // takes effect at crt_build_accel
    if (!std::strcmp(name, "wf_finish_at")) { c->wf_finish_at = (uint32_t)std::max<int64_t>(0, value); return CRT_OK; }
    if (!std::strcmp(name, "wf_flush_at")) { c->wf_flush_at = (uint32_t)std::max<int64_t>(0, value); return CRT_OK; }
    if (!std::strcmp(name, "wf_side_ppw")) { c->wf_side_ppw = (uint32_t)std::min<int64_t>(64, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_flush_ppw")) { c->wf_flush_ppw = (uint32_t)std::min<int64_t>(64, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_ring")) { c->wf_ring = (int)std::min<int64_t>(kWfRing, std::max<int64_t>(2, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_chunk")) { c->wf_chunk = (int)std::min<int64_t>(16, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_park_its")) { c->wf_park_its = (int)std::max<int64_t>(0, value); return CRT_OK; }
    if (!std::strcmp(name, "wf_tail_walk")) { c->wf_tail_walk = value != 0; return CRT_OK; }
    if (!std::strcmp(name, "wf_pipes")) { c->wf_pipes = (int)std::min<int64_t>(crt_ctx::kMaxPipes, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_pool")) { c->wf_pool = (uint32_t)std::max<int64_t>(0, value); return CRT_OK; }
    if (!std::strcmp(name, "wf_waves_per_cu")) { c->wf_waves_per_cu = (uint32_t)std::min<int64_t>(32, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "time_kernels")) {
        // value > 1 also creates the event pairs for that many launches now (event creation costs ~10 us apiece,
        // which would otherwise land in the region being timed)
        c->time_kernels = value != 0; c->last_trace_kernel_launches = 0;
        HIPCHK(c, hipSetDevice(c->device));
        while (value > 1 && c->kev.size() < 2 * (size_t)std::min<int64_t>(value, 1 << 20)) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreate(&e));
            c->kev.push_back(e);
        }
        return CRT_OK;
    }
    return fail(c, CRT_EINVAL, "crt_set_option: unknown option '%s'", name);
}

int crt_debug_intersect(crt_ctx *c, const float *rays, size_t n, float *out)
{
    if (!c || (!rays && n) || (!out && n)) return CRT_EINVAL;
    if (!c->have_scene || c->accel_mode < 0) return fail(c, CRT_ESTATE, "crt_debug_intersect: scene + accel required");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf<float> din, dout;
    HIPCHK(c, din.alloc(n * 8));
    hipError_t e = dout.alloc(n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(din.p, rays, n * 8 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_debug_intersect(c->sc, din.p, n, dout.p, c->accel_mode == CRT_ACCEL_NONE, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, n * 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    din.release(); dout.release();
    if (e != hipSuccess) return fail(c, CRT_EDEVICE, "crt_debug_intersect: %s", hipGetErrorString(e));
    return CRT_OK;
}

int crt_debug_math(crt_ctx *c, int fn, const float *a, const float *b, float *out, size_t n)
{
    if (!c || !a || !b || !out) return CRT_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf<float> da, db, dout;
    hipError_t e = da.alloc(n);
    if (e == hipSuccess) e = db.alloc(n);
    if (e == hipSuccess) e = dout.alloc(n);
    if (e == hipSuccess && n) e = hipMemcpyAsync(da.p, a, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(db.p, b, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_debug_math(fn, da.p, db.p, dout.p, n, c->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(out, dout.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    da.release(); db.release(); dout.release();
    if (e != hipSuccess) return fail(c, CRT_EDEVICE, "crt_debug_math: %s", hipGetErrorString(e));
    return CRT_OK;
}

}  // extern "C"
