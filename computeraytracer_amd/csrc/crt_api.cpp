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
#include <chrono>
#include <string>
#include <thread>
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
hipError_t wf_launch_gen(const WfParams &P, uint32_t it, hipStream_t s);
hipError_t wf_launch_trace(const WfParams &P, uint32_t it, uint32_t trace_blocks, hipStream_t s);
hipError_t wf_launch_finish(const WfParams &P, WfFinishSegs G, uint32_t max_paths, hipStream_t s);
hipError_t wf_launch_resolve(const WfParams &P, uint32_t last_sample, hipStream_t s);
hipError_t build_lbvh(const float *lo, const float *hi, uint32_t n, Bvh &out, hipStream_t stream);
hipError_t build_lbvh_device(const unsigned char *d_raw, uint32_t n, float hit_pad, float4 *d_prim, float4 *d_primD,
                             uint32_t *d_slot_of_index, float *d_nodes2, uint4 *d_nodes4q, LbvhDeviceResult &res, hipStream_t stream);
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
    DevBuf<unsigned char> d_raw;    // the scene's 80-byte records as uploaded (input of the all-device LBVH build)
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
    DevBuf<uchar4> d_frames;        // option "frame_ring" = F: the rgba8 frame of each of the last F samples (tile-sized each)
    uint32_t frame_ring = 0;
    std::vector<uint8_t> frame_batch;   // per ring slot: the batch id whose resolve pass wrote that frame (its ev_resolved orders a read)
    hipStream_t read_stream = nullptr;  // readbacks from the ring: they wait for the frame's own resolve pass, not for the retirement work queued behind it
    uint32_t ring_from = 1;         // the ring holds frames of samples >= this (a restored accumulator brings no frames with it)
    uint32_t resolved_upto = 0;     // samples whose resolve pass has been enqueued on the context's stream (frames <= this are in the ring / the framebuffer in stream order)
    float4 *accum_bound = nullptr;
    uchar4 *rgba_bound = nullptr;
    uint32_t sample = 0;            // samples requested so far (ComputeShader.wgsl:3 after that many frames)
    uint32_t published = 0;         // ... of which this many have been turned into batches (or run by the single-kernel form)
    uint32_t pending = 0;           // ... and this many wait to be merged with the next calls' (sample == published + pending)
    bool in_publish = false;
    int wf_cohort = 16;             // small calls are merged into batches of at least this many samples (1 = every call its own batch)

    DevBuf<unsigned long long> d_counters;
    bool counting = false;
    float last_ms = 0.0f;
    uint32_t last_launches = 0;
    bool last_timed = false;
    uint32_t spp_per_launch = 0;    // 0 = auto

    // wavefront pipeline (crt_wavefront.hip)
    int pipeline = 1;               // 1 = wavefront (default), 0 = v1 megakernel
    uint32_t wf_pool = 0;           // 0 = auto
    uint32_t wf_waves_per_cu = 0;   // persistent traversal waves per CU and pipe; 0 = auto: 13 for k_wf_trace2, 16 for k_wf_trace (the
                                    // regrouped form does more per wave and leaves the shade waves of the other pipe more of the SIMDs:
                                    // profiles/r03_ab_waves.txt)
    int num_cu = 0;
    DevBuf<float4> w_ray_o, w_ray_d, w_sh_d, w_beta, w_radiance, w_nee, w_staging[kWfRing], w_recA, w_recB;
    DevBuf<uint4> w_rng, w_misc, w_recC;
    DevBuf<float2> w_hit;
    DevBuf<uint32_t> w_vis, w_dead, w_tea;
    // up to kMaxPipes half-pools, each its own shade->trace chain on its own stream
    static constexpr int kMaxPipes = 4;
    int wf_pipes = 2;
    int wf_defer = 1;               // 1: crt_trace returns with its batch in flight; its paths finish under the next batches (or at crt_sync)
    int wf_tail_walk = 1;           // shade walks the ray lists once few paths are left
    int wf_gen_blocks = 128;        // k_wf_gen: waves per shard (64 shards)
    int wf_trace_form = 2;          // traversal kernel: 2 = k_wf_trace2 (ray ring + primitive tasks), 1 = k_wf_trace
    int wf_chunk = 1;               // iterations per status record at most
    int wf_ahead = 3;               // iterations in flight per pipe before the pump waits for a status
    int wf_ring = 32;               // batches in flight at most (2..kWfRing): bounds how many calls a bound output can lag
    int wf_pool_spp = 8;            // automatic pool size: at least this many path slots per tile pixel (within 1 M .. 24 M)
    double wf_feed = 1.0;           // pump: weight of the work the iterations in flight are expected to consume
    WfRun *run = nullptr;           // pipeline state between calls
    uint32_t wf_finish_at = 32768;  // paths of the oldest batch left (per pipe) at which they move to the side pool; 0 = never
    uint32_t wf_flush_at = 4096;    // the same for the LAST batch at crt_sync (nothing to hide its tail under); 0 = never
    uint32_t wf_side_ppw = 64, wf_flush_ppw = 4;   // k_wf_finish: paths per wave, under the next batch / at crt_sync
    DevBuf<WfCtl> w_ctl[kMaxPipes];
    DevBuf<WfWorkQ> w_wq;
    static constexpr int kStatusSlots = 64;                // status records per pipe (one per iteration in flight)
    WfStatus *h_status[kMaxPipes] = {};                    // pinned host records, written by k_wf_status ...
    WfStatus *d_status[kMaxPipes] = {};                    // ... through these device pointers
    hipEvent_t ev_status[kMaxPipes][kStatusSlots] = {};
    hipEvent_t ev_done[kMaxPipes][kStatusSlots] = {};      // after the traversal launch of that iteration
    uint32_t *h_dropped = nullptr;                         // pinned [kMaxPipes]: WfCtl::dropped after the last flush
    bool wf_host_ready = false;                            // the streams / events / pinned buffers below exist
    hipStream_t pipe_stream[kMaxPipes] = {};               // the pipes' own streams (the context's stream sets up, finishes stragglers and resolves)
    hipStream_t pub_stream = nullptr;                      // publishes a new batch's queue (waits only for what it must)
    static constexpr int kFinishStreams = 3;
    hipStream_t fin_stream[kFinishStreams] = {};           // k_wf_finish launches (lowest priority; each lasts as long as its longest path,
    hipEvent_t ev_fin[kFinishStreams] = {};                //  so consecutive ones overlap); the resolve passes wait for these events
    int fin_next = 0;
    hipEvent_t ev_fork = nullptr, ev_join[kMaxPipes] = {}, ev_pub_join[kMaxPipes] = {};
    hipEvent_t ev_evict[kMaxPipes][kWfRing] = {};          // after the shade launch of that pipe that evicts that batch id
    hipEvent_t ev_resolved[kWfRing] = {};                  // after the resolve pass of the batch that used the id last
    hipEvent_t ev_pub[kWfRing] = {};                       // after the queue reset of the batch that uses the id now
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
    char buf[2048];
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
    if (c->frame_ring) HIPCHK(c, c->d_frames.alloc(n * c->frame_ring)); else c->d_frames.release();
    c->frame_batch.assign(c->frame_ring, 0);
    if (c->frame_ring && !c->read_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->read_stream, hipStreamNonBlocking));
    return CRT_OK;
}

// (the frame ring alone: option "frame_ring" changes it without touching the accumulator)
int alloc_frames(crt_ctx *c)
{
    const size_t n = (size_t)c->tw * c->th;
    if (c->frame_ring) HIPCHK(c, c->d_frames.alloc(n * c->frame_ring)); else c->d_frames.release();
    c->frame_batch.assign(c->frame_ring, 0);
    if (c->frame_ring && !c->read_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->read_stream, hipStreamNonBlocking));
    return CRT_OK;
}

int zero_state(crt_ctx *c)
{
    size_t n = (size_t)c->tw * c->th;
    if (n) {
        HIPCHK(c, hipMemsetAsync(accum_ptr(c), 0, n * sizeof(float4), c->stream));
        HIPCHK(c, hipMemsetAsync(rgba_ptr(c), 0, n * sizeof(uchar4), c->stream));
    }
    c->sample = 0; c->published = 0; c->pending = 0; c->resolved_upto = 0; c->ring_from = 1;
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

// crt_build_accel(CRT_ACCEL_LBVH), all on the device (crt_lbvh.hip): bounds, Morton order, hierarchy, collapse to the
// 4-wide tree, quantisation and the leaf-ordered primitive records; nothing of the tree visits the host.  Returns
// CRT_OK with *done = false when the scene cannot take this path (not quantisable): the caller builds the host way.
int build_accel_on_device(crt_ctx *c, bool *done)
{
    *done = false;
    const uint32_t n = (uint32_t)c->prims.size();
    if (n < 2 || !c->quantize || c->wf_width != 4 || c->d_raw.n < (size_t)n * 80) return CRT_OK;
    HIPCHK(c, c->d_prim.alloc((size_t)n * 3));
    HIPCHK(c, c->d_primD.alloc(n));
    HIPCHK(c, c->d_slot_of_index.alloc(n));
    HIPCHK(c, c->d_nodes.alloc((size_t)(n - 1) * 4));
    HIPCHK(c, c->d_nodes4q.alloc((size_t)(n - 1) * 4));
    HIPCHK(c, c->d_nodes4.alloc(8));
    LbvhDeviceResult res;
    const hipError_t e = build_lbvh_device(c->d_raw.p, n, c->sc.hit_pad, c->d_prim.p, c->d_primD.p, c->d_slot_of_index.p,
                                           (float *)c->d_nodes.p, c->d_nodes4q.p, res, c->stream);
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? CRT_ENOMEM : CRT_EDEVICE, "crt_build_accel: GPU LBVH build: %s", hipGetErrorString(e));
    if (!res.quantised) return CRT_OK;
    // the host keeps the tree's statistics only
    c->bvh = Bvh(); c->bvh4 = Bvh4(); c->bvh4q = Bvh4Q(); c->bvh8q = Bvh8Q();
    c->bvh.root = 0; c->bvh.n_inner = n - 1; c->bvh.n_leaves = n; c->bvh.max_depth = res.max_depth;
    c->bvh4.root = 0; c->bvh4.n_inner = res.n_nodes4;
    c->bvh4q.ok = true;
    for (int a = 0; a < 3; a++) { c->bvh4q.base[a] = res.qbase[a]; c->bvh4q.scale[a] = res.qscale[a]; c->sc.qbase[a] = res.qbase[a]; c->sc.qscale[a] = res.qscale[a]; }
    c->accel_builder = 1;
    c->sc.prim = c->d_prim.p; c->sc.primD = c->d_primD.p; c->sc.slot_of_index = c->d_slot_of_index.p;
    c->sc.nodes = c->d_nodes.p; c->sc.root = 0;
    c->sc.nodes4 = c->d_nodes4.p; c->sc.root4 = 0; c->sc.n_nodes4 = res.n_nodes4;
    c->sc.nodes4q = c->d_nodes4q.p;
    c->sc.nodes8q = nullptr; c->sc.root8 = -1;
    c->sc.nprim = n;
    c->sc.npatch = 0;
    for (const HostPrim &hp_ : c->prims) c->sc.npatch += hp_.category == 0u ? 1u : 0u;
    c->accel_mode = CRT_ACCEL_BVH2;
    *done = true;
    return CRT_OK;
}

// Builds the device primitive arrays in `order` and (for BVH2) the node array.
int upload_geometry(crt_ctx *c, int mode)
{
    const uint32_t n = (uint32_t)c->prims.size();
    const float pad = c->sc.hit_pad;
    if (mode == CRT_ACCEL_BVH2 && c->want_lbvh) {
        bool done = false;
        int rc = build_accel_on_device(c, &done);
        if (rc || done) return rc;
    }
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
    c->sc.npatch = 0;
    for (const HostPrim &hp_ : c->prims) c->sc.npatch += hp_.category == 0u ? 1u : 0u;
    c->accel_mode = mode;
    return CRT_OK;
}


// ---------------------------------------------------------------- wavefront driver
//
// The pool (crt_wavefront.hip) is a steady-state machine: every iteration of a pipe is one shade launch (advance
// every path slot by one bounce, re-arm dead slots from the work queues) and one traversal launch.  The host's
// part is to keep it fed and to retire finished batches:
//
//   publish   a crt_trace call becomes a BATCH: a work queue, a staging buffer and side pools of its own, indexed by
//             the batch id that a path carries in its flags.  Up to `ring` batches are in flight.
//   pump      enqueue as many iterations as the published work needs -- by an estimate of how many work items one
//             iteration consumes, corrected by every status that comes back -- and return.  The call does NOT wait
//             for its work: statuses are polled (hipEventQuery), the only blocking waits are back-pressure (the
//             ring of batches or of status buffers is full).  A loop of 1-spp calls (the reference's frame loop,
//             main.js:597-611) therefore runs the pool exactly like one large batch does.
//   retire    batches resolve in order (the accumulator is summed in sample order).  A batch whose queue is dry
//             for every pipe and of which few paths are left has those EVICTED by the pipes' next shade launch into
//             side pools; k_wf_finish runs them to their end (one launch for all pipes and every batch that is ready)
//             and k_wf_resolve adds the batch to the accumulator -- on the context's stream, under the pool's work.
//   flush     crt_sync and every call that reads or changes state: run everything to its end.
//
//   cohorts   small calls are merged: crt_trace only notes their samples, which become one batch once wf_cohort of
//             them have come together (wf_publish_pending) -- a batch of many samples per pixel keeps the paths in
//             flight inside a band of the image.
//
// What the driver needs to know about an iteration (queue cursors, rays listed, paths alive per batch) is written
// into a pinned host record by the first wave of the NEXT iteration's shade launch (write_status) and polled here.
constexpr int kStatusRing = crt_ctx::kStatusSlots;

uint32_t wf_waves(const crt_ctx *c) { return c->wf_waves_per_cu ? c->wf_waves_per_cu : (c->wf_trace_form == 2 && c->bvh4q.ok && !c->bvh8q.ok) ? 13u : 16u; }

int wf_ensure(crt_ctx *c, size_t P, size_t staging_elems, size_t list_elems, uint32_t ring)
{
    if (c->w_dead.n < list_elems / 8) HIPCHK(c, c->w_dead.alloc(list_elems / 8));   // dead-slot lists: one list's worth per pipe
    if (c->w_recA.n < list_elems) HIPCHK(c, c->w_recA.alloc(list_elems));      // the ray records = the ray lists
    if (c->w_recB.n < list_elems) HIPCHK(c, c->w_recB.alloc(list_elems));
    if (c->w_recC.n < list_elems) HIPCHK(c, c->w_recC.alloc(list_elems));
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
    for (uint32_t b = 0; b < ring; b++)
        if (c->w_staging[b].n < staging_elems) HIPCHK(c, c->w_staging[b].alloc(staging_elems));
    if (c->w_tea.n < (size_t)c->tw * c->th) HIPCHK(c, c->w_tea.alloc((size_t)c->tw * c->th));
    if (!c->wf_host_ready) {
        if (!c->w_wq.p) {
            HIPCHK(c, c->w_wq.alloc(kWfRing));
            HIPCHK(c, hipMemset(c->w_wq.p, 0, kWfRing * sizeof(WfWorkQ)));
        }
        if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        if (!c->pub_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->pub_stream, hipStreamNonBlocking));
        for (int f = 0; f < crt_ctx::kFinishStreams; f++) {
            if (!c->fin_stream[f]) {
                int least = 0, greatest = 0;
                HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIPCHK(c, hipStreamCreateWithPriority(&c->fin_stream[f], hipStreamNonBlocking, least));
            }
            if (!c->ev_fin[f]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fin[f], hipEventDisableTiming));
        }
        if (!c->h_dropped) {
            HIPCHK(c, hipHostMalloc((void **)&c->h_dropped, crt_ctx::kMaxPipes * sizeof(uint32_t), hipHostMallocDefault));
            std::memset(c->h_dropped, 0, crt_ctx::kMaxPipes * sizeof(uint32_t));
        }
        for (uint32_t b = 0; b < kWfRing; b++) {
            if (!c->ev_resolved[b]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_resolved[b], hipEventDisableTiming));
            if (!c->ev_pub[b]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_pub[b], hipEventDisableTiming));
        }
        for (int p = 0; p < crt_ctx::kMaxPipes; p++) {
            if (!c->w_ctl[p].p) {
                HIPCHK(c, c->w_ctl[p].alloc(1));
                HIPCHK(c, hipMemset(c->w_ctl[p].p, 0, sizeof(WfCtl)));
            }
            if (!c->h_status[p]) {
                // coherent pinned host memory that the shade kernel's first wave writes directly (write_status)
                HIPCHK(c, hipHostMalloc((void **)&c->h_status[p], kStatusRing * sizeof(WfStatus), hipHostMallocMapped | hipHostMallocCoherent));
                std::memset(c->h_status[p], 0, kStatusRing * sizeof(WfStatus));
                HIPCHK(c, hipHostGetDevicePointer((void **)&c->d_status[p], c->h_status[p], 0));
            }
            for (int k = 0; k < kStatusRing; k++) {
                if (!c->ev_status[p][k]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_status[p][k], hipEventDisableTiming));
                if (!c->ev_done[p][k]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_done[p][k], hipEventDisableTiming));
            }
            if (!c->ev_join[p]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[p], hipEventDisableTiming));
            if (!c->ev_pub_join[p]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_pub_join[p], hipEventDisableTiming));
            for (uint32_t b = 0; b < kWfRing; b++)
                if (!c->ev_evict[p][b]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_evict[p][b], hipEventDisableTiming));
        }
        c->wf_host_ready = true;
    }
    if (c->num_cu == 0) {
        hipDeviceProp_t prop;
        HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
        c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    {   // deep-stack overflow area: kWfOverflowLevels levels beyond the LDS part for every resident traversal lane
        const size_t lanes = (size_t)c->num_cu * wf_waves(c) * 64u * (size_t)std::max(1, c->wf_pipes);
        if (c->w_overflow.n < lanes * kWfOverflowLevels) HIPCHK(c, c->w_overflow.alloc(lanes * kWfOverflowLevels));
    }
    return CRT_OK;
}

// One shade->trace chain over its share of the pool.
struct WfPipe {
    WfParams W{};
    hipStream_t stream = nullptr;
    uint32_t it = 0;                // the next iteration to enqueue (numbers are never reused: a run starts where the last one
    uint32_t it_first = 0;          // ended, so a late status write of the last run cannot be taken for one of this run)
    uint32_t it_confirmed = 0;      // iterations < it_confirmed are known to have completed (a status of them was read)
    uint32_t it_done = 0;           // iterations < it_done are known to have completed (their traversal launch's event has)
    uint32_t chunk = 2, tail_bound = 0, blocks_now = 0;
    // Status records: iteration i's is written by the shade launch of iteration i + 1 into slot i % kStatusRing and
    // is pending while it_confirmed <= i < it.
    bool st_counted[kStatusRing] = {};      // that iteration's shade launch counted the alive paths per batch
    bool done = false;              // flush: nothing more is enqueued for this pipe (drained or emptied by eviction)
    bool any = false;               // a status has been read since the newest batch began
    unsigned long long rays = 0;    // from the last status: rays listed by its last iteration
    uint32_t bound = 0;             // ... and the most rays one shard listed
    uint32_t alive[kWfRing] = {};   // per batch id, from this pipe's last status that counted: paths alive in its pool
    bool alive_valid[kWfRing] = {};
    bool dry[kWfRing] = {};         // a status of this pipe saw that batch's queue empty (its OWN view: its count of
                                    // alive paths only bounds the future once no more such paths can start here)
    uint32_t evict_next = 0;        // evict_mask for the next shade launch
};

// A batch of samples whose paths are (or may still be) in flight.
struct WfBatch {
    uint32_t n = 0, last_sample = 0, id = 0;
    uint32_t from_it[crt_ctx::kMaxPipes] = {};   // per pipe: chunks enqueued from this iteration on know the batch
    bool ready = true;              // its queue has been reset on the device (ev_pub[id] seen complete): launches may list it
    uint32_t evict_bound = 0;       // the most paths one pipe held when the eviction was decided (sizes the finish launch)
    bool evicting = false;          // its last paths are being moved to the side pools (or none are left)
    uint32_t need_mask = 0;         // pipes whose next shade launch evicts ...
    uint32_t launched_mask = 0;     // ... and those that have enqueued it (ev_evict[p][id] recorded)
};

struct WfRun {
    bool live = false;              // the pipes are forked and hold (or may hold) paths
    int K = 0;
    uint32_t P = 0, Pp = 0, list_cap = 0, trace_blocks = 0, ring = 4;
    uint32_t seg_wps[kWfRing] = {};             // per batch id: work items per shard / in total
    unsigned long long seg_total[kWfRing] = {};
    WfPipe pipes[crt_ctx::kMaxPipes];
    std::vector<WfBatch> open;                  // unresolved batches, oldest first; back() = the newest
    bool queue_left[kWfRing] = {};              // that batch's queue still holds work (latest knowledge of any pipe)
    unsigned long long consumed[kWfRing] = {};  // work items taken from it (latest knowledge)
    unsigned long long consumed_total = 0;      // ... summed over all batches since the pool started
    bool work_left = false;                     // any open batch's queue holds work
    double per_it = 0.0;                        // work items one iteration of one pipe consumes while work is there (estimate)
    unsigned long long rate_consumed = 0;       // sample point of that estimate
    unsigned long long rate_its = 0;
    uint32_t listed_until[kWfRing][crt_ctx::kMaxPipes] = {};   // per id and pipe: launches of iterations < this may look at that queue
    bool resolved_recorded[kWfRing] = {};       // ev_resolved[id] has been recorded since the pool started
    bool all_evicting = false;                  // flush: everything alive was sent to the side pools
    int poll_next = 0;                          // round robin over the pipes for blocking waits
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
    // pool: between 1 M and 24 M slots, about a quarter of a LARGE batch's paths and at least wf_pool_spp (8) per tile
    // pixel; a stream of small batches shares the pool, so it is sized by the tile, not by one call's samples.  (8 M was
    // the cap while the pool's streams still washed the scene out of the L2 with every launch; with them non-temporal a
    // launch costs what its rays cost plus a fixed ramp-up, tail and gap, and larger launches amortise those:
    // profiles/r02_pool_sweep.txt, r02_pool_bench.txt -- 8 / 16 / 24 M slots: 63.9 / 61.4 / 60.5 ms per 64-spp step,
    // 1.035 / 1.001 / 1.007 ms per 1-spp step.)
    const unsigned long long per_pixel = std::max<unsigned long long>(g.work_total / 4u, (unsigned long long)g.npix_padded * (unsigned long long)c->wf_pool_spp);
    uint32_t P = c->wf_pool ? c->wf_pool : (uint32_t)std::min<unsigned long long>(3u << 23, std::max<unsigned long long>(1u << 20, per_pixel));
    // (a pool may exceed one batch's work: several batches share it, but never more than the ring holds)
    const unsigned long long most = g.work_total * (unsigned long long)std::max(1u, std::min<uint32_t>(c->wf_ring, kWfRing) - 1u);
    if ((unsigned long long)P > most) P = (uint32_t)most;
    // Two (or more) half-pools on separate streams: one half's shade pass (an HBM stream) overlaps
    // the other half's traversal (latency-bound).  Small pools keep one pipe.
    int K = std::max(1, std::min(c->wf_pipes, (int)crt_ctx::kMaxPipes));
    if (P < (1u << 19)) K = 1;
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
    R.frames = c->d_frames.p; R.frame_ring = c->frame_ring;
    HIPCHK(c, wf_launch_resolve(R, b.last_sample, c->stream));
    c->resolved_upto = b.last_sample;
    if (c->frame_ring && c->frame_batch.size() == c->frame_ring)
        for (uint32_t k = 0; k < b.n && k < c->frame_ring; k++) c->frame_batch[(b.last_sample - 1u - k) % c->frame_ring] = (uint8_t)b.id;
    HIPCHK(c, hipEventRecord(c->ev_resolved[b.id], c->stream));    // the id's queue, staging buffer and side pools are free after this
    r.resolved_recorded[b.id] = true;
    c->last_launches++;
    return CRT_OK;
}

// The host has seen the batch's queue reset complete: from here on launches may list the queue, and the status
// records of the iterations enqueued from here on describe THIS batch's queue (earlier ones may have looked at the
// cursors of the id's previous user).
bool wf_check_ready(crt_ctx *c, WfBatch &b, bool wait)
{
    if (b.ready) return true;
    if (wait) { if (hipEventSynchronize(c->ev_pub[b.id]) != hipSuccess) return false; }
    else if (hipEventQuery(c->ev_pub[b.id]) != hipSuccess) return false;
    b.ready = true;
    for (int p = 0; p < c->run->K; p++) b.from_it[p] = c->run->pipes[p].it;
    return true;
}

// Which queues the next launches re-arm from: the open batches whose queue still holds work, oldest first.
void wf_set_queues(crt_ctx *c, WfPipe &pp)
{
    WfRun &r = *c->run;
    uint32_t order[kWfRing], n = 0;
    for (WfBatch &b : r.open)
        if (wf_check_ready(c, b, false) && r.queue_left[b.id] && n < kWfRing) order[n++] = b.id;
    if (n == 0) order[n++] = r.open.empty() ? 0u : r.open.back().id;       // (all dry: any valid entry)
    for (uint32_t k = 0; k < kWfRing; k++) pp.W.seg_order[k] = order[k < n ? k : n - 1];
    pp.W.seg_n = n;
}

int wf_retire_front(crt_ctx *c);
std::string wf_state(crt_ctx *c);
constexpr double kWfStallMs = 30000.0;     // a driver loop that makes no progress for this long gives up with CRT_EDEVICE

bool wf_debug() { static const bool on = getenv("CRT_DEBUG") != nullptr; return on; }
double wf_now_ms()
{
    static const auto t_ref = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ref).count();
}


// Enqueue `iters` iterations of pipe p.  The shade launch of iteration i also writes iteration i - 1's status record.
int wf_enqueue(crt_ctx *c, int p, uint32_t iters)
{
    WfRun &r = *c->run;
    WfPipe &pp = r.pipes[p];
    if (pp.it - pp.it_confirmed + iters >= (uint32_t)kStatusRing || iters == 0) return fail(c, CRT_EDEVICE, "wavefront driver: status ring overrun");
    pp.W.tail_bound = pp.tail_bound;
    pp.W.count_alive = r.open.size() > 1 ? 1u : 0u;
    wf_set_queues(c, pp);
    // Dead slots are listed by the shade launch and re-armed by a k_wf_gen launch behind it while a listed queue may
    // hold work (the host's view lags the device's: a launch too many finds the queues dry, a launch too few leaves
    // the slots dead for one more iteration).
    pp.W.rearm = (pp.tail_bound == 0u && r.work_left) ? 1u : 0u;
    const bool evicted = pp.evict_next != 0;
    for (uint32_t k = 0; k < iters; k++, pp.it++) {
        pp.W.evict_mask = pp.evict_next;
        pp.W.status_out = pp.it > pp.it_first ? c->d_status[p] + (pp.it - 1u) % kStatusRing : nullptr;
        HIPCHK(c, wf_launch_shade(pp.W, pp.it, pp.stream));
        if (pp.it > pp.it_first) HIPCHK(c, hipEventRecord(c->ev_status[p][(pp.it - 1u) % kStatusRing], pp.stream));   // (blocking waits fall back on it)
        pp.st_counted[pp.it % kStatusRing] = pp.W.count_alive != 0;
        if (pp.W.rearm) { HIPCHK(c, wf_launch_gen(pp.W, pp.it, pp.stream)); c->last_launches++; }
        if (pp.evict_next) {                                     // k_wf_finish may start once this launch is through
            for (WfBatch &b : r.open)
                if ((pp.evict_next >> b.id) & 1u) {
                    HIPCHK(c, hipEventRecord(c->ev_evict[p][b.id], pp.stream));
                    b.launched_mask |= 1u << p;
                }
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
        HIPCHK(c, hipEventRecord(c->ev_done[p][pp.it % kStatusRing], pp.stream));
        c->last_launches += 2;
        c->last_iterations++;
    }
    for (uint32_t k = 0; k < pp.W.seg_n; k++) r.listed_until[pp.W.seg_order[k]][p] = pp.it;
    return evicted && !r.all_evicting ? wf_retire_front(c) : CRT_OK;   // a batch may have become ready for its finish pass
}

// k_wf_finish for the leading batches whose evicting launches are all enqueued, then their resolve passes -- on
// the context's stream, which has nothing else to do while the pipes work.  One finish launch covers every pipe
// of every such batch: its duration is that of the longest path in it (one lane per path, a bounce after the
// other), so batches that are ready together cost one such tail, not one each.
int wf_retire_front(crt_ctx *c)
{
    WfRun &r = *c->run;
    size_t n = 0;
    while (n < r.open.size() && r.open[n].evicting && (r.open[n].need_mask & ~r.open[n].launched_mask) == 0) n++;
    if (n == 0) return CRT_OK;
    // The finish launch goes to one of a few low-priority streams in turn (a launch lasts as long as its longest path,
    // a few milliseconds on S2 whatever the number of paths, so consecutive launches must overlap or the retirement
    // of small batches is bound by that latency); the resolve passes wait for it on the context's stream.
    const int f = c->fin_next;
    c->fin_next = (c->fin_next + 1) % crt_ctx::kFinishStreams;
    hipStream_t fs = c->fin_stream[f];
    WfFinishSegs G{};
    bool any = false;
    uint32_t bound = 1;                                         // paths per (batch, pipe) at most: alive slots only shrink once a queue is dry
    for (size_t i = 0; i < n; i++) bound = std::max(bound, r.open[i].evict_bound);
    auto launch = [&]() -> int {
        if (G.n == 0) return CRT_OK;
        WfParams F = r.pipes[0].W;
        F.tail_bound = c->wf_side_ppw;
        HIPCHK(c, wf_launch_finish(F, G, bound, fs));
        c->last_launches++;
        G.n = 0;
        any = true;
        return CRT_OK;
    };
    for (size_t i = 0; i < n; i++) {
        const WfBatch &b = r.open[i];
        for (int p = 0; p < r.K; p++) {
            if (!((b.need_mask >> p) & 1u)) continue;
            HIPCHK(c, hipStreamWaitEvent(fs, c->ev_evict[p][b.id], 0));
            if (G.n == kWfFinishSegs) { int rc = launch(); if (rc) return rc; }
            G.ctl[G.n] = r.pipes[p].W.ctl; G.base[G.n] = r.pipes[p].W.side_base[b.id]; G.batch[G.n] = b.id; G.n++;
        }
    }
    { int rc = launch(); if (rc) return rc; }
    if (any) {
        HIPCHK(c, hipEventRecord(c->ev_fin[f], fs));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_fin[f], 0));
    }
    for (size_t i = 0; i < n; i++) { int rc = wf_resolve_batch(c, r.open[i]); if (rc) return rc; }   // in order
    r.open.erase(r.open.begin(), r.open.begin() + (long)n);
    return CRT_OK;
}

// Retirement decisions after a status: the oldest batches (never the newest -- flush does that one) whose queue is
// dry for every pipe and of which few paths are left (alive slots only shrink once the queue is dry, so they still
// fit when the launch runs) have them evicted by every pipe's next shade launch; none left: nothing to evict.
int wf_retire(crt_ctx *c)
{
    WfRun &r = *c->run;
    if (r.all_evicting) return CRT_OK;
    const unsigned long long evict_at = std::min<unsigned long long>(c->wf_finish_at, kWfSideCap);
    for (size_t i = 0; i + 1 < r.open.size(); i++) {
        WfBatch &b = r.open[i];
        if (b.evicting) continue;
        bool ready = true;
        unsigned long long tot = 0, mx = 0;
        uint32_t mask = 0;
        for (int p = 0; p < r.K; p++) {
            const WfPipe &pp = r.pipes[p];
            if (pp.done) continue;                               // (a drained pipe holds no path at all)
            ready = ready && pp.alive_valid[b.id] && pp.dry[b.id];
            tot += pp.alive[b.id]; mx = std::max<unsigned long long>(mx, pp.alive[b.id]);
            if (pp.alive[b.id]) mask |= 1u << p;
        }
        if (!ready) break;                                       // in order
        if (tot != 0 && !(mx <= kWfSideCap && tot <= evict_at * (unsigned)r.K)) break;
        b.evicting = true; b.need_mask = mask; b.launched_mask = 0; b.evict_bound = (uint32_t)mx;
        for (int p = 0; p < r.K; p++) if ((mask >> p) & 1u) r.pipes[p].evict_next |= 1u << b.id;
    }
    return wf_retire_front(c);
}

// One status record of pipe p has arrived: fold it into the driver's view.
int wf_process_status(crt_ctx *c, int p)
{
    WfRun &r = *c->run;
    WfPipe &pp = r.pipes[p];
    const int slot = (int)(pp.it_confirmed % kStatusRing);
    const WfStatus st = c->h_status[p][slot];                   // (the caller has seen it_end == it_confirmed + 1, with acquire)
    const bool counted = pp.st_counted[slot];
    if (st.it_end != pp.it_confirmed + 1u) return fail(c, CRT_EDEVICE, "wavefront driver: status record out of order (pipe %d: %u, expected %u)", p, st.it_end, pp.it_confirmed + 1u);
    if (st.dropped) return fail(c, CRT_EDEVICE, "wavefront pipeline: a capacity guard dropped %u paths (pipe %d)", st.dropped, p);
    pp.it_confirmed = st.it_end;
    for (const WfBatch &b : r.open) {
        if (st.it_end <= b.from_it[p]) continue;                 // from before that batch began: knows nothing about it
        const uint32_t id = b.id;
        if (!st.left[id]) { pp.dry[id] = true; r.queue_left[id] = false; }      // (monotone within a batch)
        if (counted) { pp.alive[id] = st.alive[id]; pp.alive_valid[id] = true; }
        if (st.consumed[id] > r.consumed[id]) { r.consumed_total += st.consumed[id] - r.consumed[id]; r.consumed[id] = st.consumed[id]; }
    }
    r.work_left = false;
    for (const WfBatch &b : r.open) r.work_left = r.work_left || r.queue_left[b.id];
    pp.rays = st.rays; pp.bound = st.bound;
    if (!r.open.empty() && st.it_end > r.open.back().from_it[p]) pp.any = true;
    // work one iteration of one pipe consumes: sampled over intervals at whose end work was still there
    unsigned long long its = 0;
    for (int q = 0; q < r.K; q++) its += r.pipes[q].it_confirmed;
    if (its > r.rate_its) {
        if (r.work_left && r.consumed_total > r.rate_consumed) {
            const double sample = (double)(r.consumed_total - r.rate_consumed) / (double)(its - r.rate_its);
            r.per_it = 0.5 * r.per_it + 0.5 * sample;
        }
        r.rate_its = its; r.rate_consumed = r.consumed_total;
    }
    if (getenv("CRT_DEBUG")) {
        fprintf(stderr, "[crt  %8.2f] pipe %d it %u rays %llu open %zu work_left %d per_it %.0f alive", wf_now_ms(), p, st.it_end, st.rays, r.open.size(), (int)r.work_left, r.per_it);
        for (const WfBatch &b : r.open) fprintf(stderr, " %u:%u%s", b.id, pp.alive[b.id], b.evicting ? "e" : pp.dry[b.id] ? "d" : "");
        fprintf(stderr, "\n");
    }
    return wf_retire(c);
}

// Look at pipe p's oldest pending status record; block = wait for it.  *got says whether one was processed.
int wf_poll(crt_ctx *c, int p, bool block, bool *got)
{
    WfPipe &pp = c->run->pipes[p];
    if (got) *got = false;
    if (pp.it_confirmed >= pp.it) return CRT_OK;
    const uint32_t want = pp.it_confirmed + 1u;
    const int slot = (int)(pp.it_confirmed % kStatusRing);
    const uint32_t *flag = &c->h_status[p][slot].it_end;
    auto ready = [&]() { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == want; };
    if (!ready()) {
        if (!block) return CRT_OK;
        // the record is written by the NEXT iteration's shade launch: make sure there is one
        if (pp.it == want) { int rc = wf_enqueue(c, p, 1); if (rc) return rc; }
        for (int spin = 0; spin < 2000 && !ready(); spin++) std::this_thread::yield();
        if (!ready()) {
            HIPCHK(c, hipEventSynchronize(c->ev_status[p][slot]));
            if (!ready()) return fail(c, CRT_EDEVICE, "wavefront driver: status record of pipe %d iteration %u did not arrive", p, want - 1u);
        }
    }
    if (got) *got = true;
    return wf_process_status(c, p);
}

int wf_poll_all(crt_ctx *c)
{
    WfRun &r = *c->run;
    for (int p = 0; p < r.K; p++)
        for (;;) {
            bool got = false;
            int rc = wf_poll(c, p, false, &got);
            if (rc) return rc;
            if (!got) break;
        }
    return CRT_OK;
}

// Block until some status arrives (enqueueing iterations first where nothing is in flight).
int wf_wait_progress(crt_ctx *c)
{
    WfRun &r = *c->run;
    for (int p = 0; p < r.K; p++)
        if (!r.pipes[p].done && r.pipes[p].it - r.pipes[p].it_confirmed < 2u) { int rc = wf_enqueue(c, p, r.pipes[p].chunk); if (rc) return rc; }
    for (int k = 0; k < r.K; k++) {
        const int p = (r.poll_next + k) % r.K;
        if (r.pipes[p].it == r.pipes[p].it_confirmed) continue;
        r.poll_next = (p + 1) % r.K;
        return wf_poll(c, p, true, nullptr);
    }
    return CRT_OK;
}

// Iterations of pipe p that are enqueued and have not completed.  (An iteration's status record only arrives with the
// NEXT iteration's shade launch, so a pipe that has run out of launches would look busy for ever by the records alone.)
uint32_t wf_in_flight(crt_ctx *c, int p)
{
    WfPipe &pp = c->run->pipes[p];
    if (pp.it_done < pp.it_confirmed) pp.it_done = pp.it_confirmed;
    while (pp.it_done < pp.it && hipEventQuery(c->ev_done[p][pp.it_done % kStatusRing]) == hipSuccess) pp.it_done++;
    return pp.it - pp.it_done;
}

// The driver's state in a line (for the error message of a loop that does not make progress).
std::string wf_state(crt_ctx *c)
{
    WfRun &r = *c->run;
    char buf[256];
    std::string out;
    snprintf(buf, sizeof buf, "K %d ring %u open %zu work_left %d per_it %.0f all_evicting %d |", r.K, r.ring, r.open.size(), (int)r.work_left, r.per_it, (int)r.all_evicting);
    out += buf;
    for (int p = 0; p < r.K; p++) {
        const WfPipe &pp = r.pipes[p];
        snprintf(buf, sizeof buf, " pipe %d: it %u done %u confirmed %u rays %llu any %d finished %d evict %x |", p, pp.it, pp.it_done, pp.it_confirmed, pp.rays, (int)pp.any, (int)pp.done, pp.evict_next);
        out += buf;
    }
    size_t shown = 0;
    for (const WfBatch &b : r.open) {
        if (shown++ >= 3) break;
        snprintf(buf, sizeof buf, " batch %u: left %d ready %d evicting %d need %x launched %x", b.id, (int)r.queue_left[b.id], (int)b.ready, (int)b.evicting, b.need_mask, b.launched_mask);
        out += buf;
        for (int p = 0; p < r.K; p++) {
            snprintf(buf, sizeof buf, " [p%d alive %u valid %d dry %d from %u]", p, r.pipes[p].alive[b.id], (int)r.pipes[p].alive_valid[b.id], (int)r.pipes[p].dry[b.id], b.from_it[p]);
            out += buf;
        }
        out += ";";
    }
    return out;
}

// Is there room for another batch: a free batch id whose previous user's finish / resolve passes have COMPLETED (the
// new batch's queue reset waits for them on the device; a reset that waits stalls every pipe behind it).
bool wf_has_room(crt_ctx *c)
{
    WfRun &r = *c->run;
    if (r.open.size() >= (size_t)r.ring) return false;
    const uint32_t id = r.open.empty() ? 0u : (r.open.back().id + 1u) % r.ring;
    if (!r.resolved_recorded[id]) return true;
    return hipEventQuery(c->ev_resolved[id]) == hipSuccess;
}

// ---- Invariants of the driver (publish / pump / retire), kept next to the loop that depends on all of them ----
//  I1  Iteration numbers of a pipe never repeat: a run starts 2 * kStatusRing behind the previous run's last number, so a
//      late status write of an earlier run can never match the `it_end` a poll is waiting for.
//  I2  Iteration i's status record is written by the FIRST wave of shade launch i + 1 (slot i % kStatusRing) and is
//      pending while it_confirmed <= i < it; wf_enqueue refuses to run more than kStatusRing - 1 ahead of it_confirmed, so
//      a slot is never rewritten before it has been read.  A poll accepts a slot only when its it_end is exactly
//      it_confirmed + 1 (acquire load of the word the device stores last, behind a system-scope fence).
//  I3  What a status says about a queue is FINAL for that launch: since round 3 nothing takes work inside a shade launch
//      (k_wf_gen does, between the shade launches), so cursors (`left`, `consumed`) and alive counts of one record are
//      one consistent cut -- taken after gen(i) has completed and before gen(i + 1) starts.
//  I4  A batch's statuses count only from launches enqueued after the HOST has seen its queue reset complete
//      (wf_check_ready sets from_it then; earlier launches may have read the id's previous extent).  Until then the
//      queue is not listed (wf_set_queues) and no launch can take its work.
//  I5  `dry[id]` of a pipe is monotone within a batch, and once a pipe has seen the queue dry no path of the batch can
//      start in that pipe any more (cursors only grow): alive[id] of that record bounds the pipe's paths from then on.
//      wf_retire evicts only when EVERY pipe's record says dry and the counts fit the side pools (k_wf_finish reports,
//      through WfCtl::dropped, if they did not).
//  I6  Batches resolve in publication order (the accumulator is summed in sample order); a batch id is reused only after
//      ev_resolved[id] of its previous user has COMPLETED (wf_has_room), and the reset of the id's queue additionally waits,
//      on the device, for launches that still list the old queue (listed_until) and for that resolve.
//  I7  Staging buffers, pool arrays and lists are never reallocated while r.live (wf_trace_batch flushes first).
//  I8  Every loop below makes progress or blocks on something the device will complete: a launch is always enqueued
//      behind the status being waited for (the record is written by the NEXT launch), back-pressure waits are on events
//      recorded behind enqueued work, and a wall-clock watchdog turns a violated assumption into CRT_EDEVICE + state dump.
//  I9  A flush returns with its finish / resolve passes still QUEUED on the context's stream (only crt_sync and the reads
//      wait for them).  Everything a new run starts is therefore ordered behind that stream: the pool's set-up runs on it,
//      the pipes AND the publishing stream wait for its fork event.  (I6's device-side waits are per run -- resolved_recorded
//      starts afresh -- so without the fork wait the reset of the new run's second batch could zero the side counters
//      under the previous run's k_wf_finish: round 3's lost-paths defect, test_flush_without_host_sync_then_quick_batches.)
//      WfCtl::dropped is never reset by a set-up, only once the host has reported it.
//
// Feed the pool: enqueue the iterations the published work needs (see the head of this section).  for_room = false:
// return once they are enqueued (the call does not wait for its work); for_room = true: keep feeding and reading
// statuses until there is room for another batch (back-pressure of a caller that publishes faster than the pool works).
int wf_pump(crt_ctx *c, bool for_room)
{
    WfRun &r = *c->run;
    const uint32_t max_ahead = (uint32_t)std::max(c->wf_ahead, c->wf_chunk + 1);   // iterations in flight per pipe before the driver waits
    const double t_start = wf_now_ms();
    for (int guard = 0; guard < 4000000; guard++) {
        if ((guard & 63) == 63 && wf_now_ms() - t_start > kWfStallMs) return fail(c, CRT_EDEVICE, "wavefront driver: pump stalled (%s)", wf_state(c).c_str());
        int rc = wf_poll_all(c);
        if (rc) return rc;
        if (for_room && wf_has_room(c)) return CRT_OK;
        unsigned long long backlog = 0, inflight = 0;
        for (const WfBatch &b : r.open)
            if (r.queue_left[b.id] && r.seg_total[b.id] > r.consumed[b.id]) backlog += r.seg_total[b.id] - r.consumed[b.id];
        uint32_t fl[crt_ctx::kMaxPipes];
        for (int p = 0; p < r.K; p++) { fl[p] = wf_in_flight(c, p); inflight += fl[p]; }
        const double per_it = std::max(r.per_it, 1024.0);
        const double need = (double)backlog - per_it * (double)inflight * c->wf_feed;
        if (!(need > 0.0)) {
            if (!for_room) return CRT_OK;
            // nothing to feed, but the oldest batch has yet to retire: its last paths need iterations (or only its
            // finish / resolve passes are still running on the device)
            if (wf_debug()) fprintf(stderr, "[pump %8.2f] no room and nothing to feed (open %zu)\n", wf_now_ms(), r.open.size());
            if (r.open.size() < (size_t)r.ring) {
                const uint32_t id = (r.open.back().id + 1u) % r.ring;
                HIPCHK(c, hipEventSynchronize(c->ev_resolved[id]));
                r.resolved_recorded[id] = false;                 // (complete: nothing to wait for any more)
                continue;
            }
            rc = wf_wait_progress(c);
            if (rc) return rc;
            continue;
        }
        // the pipe with the fewest iterations in flight takes the next chunk
        int p = 0;
        for (int q = 1; q < r.K; q++) if (fl[q] < fl[p]) p = q;
        WfPipe &pp = r.pipes[p];
        if (pp.it - pp.it_confirmed + (uint32_t)c->wf_chunk >= (uint32_t)kStatusRing - 1u) {
            rc = wf_poll(c, p, true, nullptr);                  // (out of status slots: wait for this pipe's oldest record)
            if (rc) return rc;
            continue;
        }
        if (fl[p] >= max_ahead) {                                // back-pressure: wait for this pipe's oldest iteration
            const double t0 = wf_debug() ? wf_now_ms() : 0.0;
            HIPCHK(c, hipEventSynchronize(c->ev_done[p][pp.it_done % kStatusRing]));
            pp.it_done++;                                        // (known now, whatever a later hipEventQuery says)
            if (wf_debug()) fprintf(stderr, "[pump %8.2f] waited %.2f ms for pipe %d (in flight %u %u, need %.0f, room %d)\n", wf_now_ms(), wf_now_ms() - t0, p, fl[0], fl[r.K - 1], need, (int)for_room);
            continue;
        }
        // (a batch whose queue reset has not been seen complete is not listed yet: wait for it rather than launch
        // iterations that cannot take its work)
        {
            bool listed = false;
            WfBatch *pending = nullptr;
            for (WfBatch &b : r.open) {
                if (!r.queue_left[b.id]) continue;
                if (wf_check_ready(c, b, false)) listed = true; else if (!pending) pending = &b;
            }
            if (!listed && pending && !wf_check_ready(c, *pending, true)) return fail(c, CRT_EDEVICE, "wavefront driver: queue reset failed");
        }
        const double want = std::ceil(need / per_it / (double)r.K);
        const uint32_t iters = (uint32_t)std::min<double>((double)c->wf_chunk, std::max(1.0, want));
        const double t0 = wf_debug() ? wf_now_ms() : 0.0;
        rc = wf_enqueue(c, p, iters);
        if (rc) return rc;
        if (wf_debug()) fprintf(stderr, "[pump %8.2f] enqueued %u on pipe %d in %.3f ms (in flight %u %u, need %.0f, open %zu)\n", wf_now_ms(), iters, p, wf_now_ms() - t0, fl[0], fl[r.K - 1], need, r.open.size());
    }
    return fail(c, CRT_EDEVICE, "wavefront driver: pump did not converge");
}

// A cheap turn of the driver for calls that publish nothing themselves (a small call that is only noted, a query of the
// latest frame): read the statuses that have arrived, and where a retirement decision waits for a pipe's next shade launch
// (evict_next), give it one -- so that batches keep retiring (finish + resolve) while a display loop runs ahead of them.
int wf_tick(crt_ctx *c)
{
    if (!c->run || !c->run->live || c->in_publish) return CRT_OK;
    WfRun &r = *c->run;
    int rc = wf_poll_all(c);
    if (rc) return rc;
    for (int p = 0; p < r.K; p++) {
        WfPipe &pp = r.pipes[p];
        if (pp.evict_next == 0 || pp.done) continue;
        if (wf_in_flight(c, p) >= (uint32_t)std::max(c->wf_ahead, c->wf_chunk + 1)) continue;
        if (pp.it - pp.it_confirmed + 1u >= (uint32_t)kStatusRing - 1u) continue;
        rc = wf_enqueue(c, p, 1);
        if (rc) return rc;
    }
    return CRT_OK;
}

// Run everything in the pool to its end and resolve every batch.
int wf_finish_all(crt_ctx *c)
{
    WfRun &r = *c->run;
    const int K = r.K;
    const unsigned long long flush_at = std::min<unsigned long long>(c->wf_flush_at, kWfSideCap);
    for (int p = 0; p < K; p++) { r.pipes[p].done = false; r.pipes[p].chunk = (uint32_t)c->wf_chunk; }
    for (WfBatch &b : r.open) if (!wf_check_ready(c, b, true)) return fail(c, CRT_EDEVICE, "wavefront driver: queue reset failed");
    r.all_evicting = false;
    int active = K;
    const double t_start = wf_now_ms();
    for (unsigned long long guard = 0; active > 0; guard++) {
        if (guard > 4000000ull || ((guard & 15) == 15 && wf_now_ms() - t_start > 4.0 * kWfStallMs))
            return fail(c, CRT_EDEVICE, "wavefront pipeline did not drain (%s)", wf_state(c).c_str());
        // one chunk is always enqueued AHEAD of the status being waited for, so the GPU never idles on the host
        // (iteration i's status record is written by the launch of iteration i + 1: three in flight = one ahead of the
        // one whose record is being waited for)
        for (int p = 0; p < K; p++)
            while (!r.pipes[p].done && r.pipes[p].it - r.pipes[p].it_confirmed < 3u) { int rc = wf_enqueue(c, p, 1); if (rc) return rc; }
        int p = -1;
        for (int k = 0; k < K && p < 0; k++) {
            const int q = (r.poll_next + k) % K;
            if (!r.pipes[q].done && r.pipes[q].it > r.pipes[q].it_confirmed) p = q;
        }
        if (p < 0) break;
        r.poll_next = (p + 1) % K;
        int rc = wf_poll(c, p, true, nullptr);
        if (rc) return rc;
        WfPipe &pp = r.pipes[p];
        // chunks shrink as the queues run dry: what is enqueued ahead of the status that shows them empty runs on a
        // nearly empty pool, and the host needs only ~20 us per launch to keep up
        if (!r.work_left) pp.chunk = 1;
        // every alive slot lists a ray: no rays and no work means this pipe is drained (another pipe's view of the
        // queues can lag; a pipe with no rays while work may be left simply keeps going)
        if (!r.work_left && pp.any && pp.rays == 0) { pp.done = true; active--; continue; }
        if (!r.work_left && pp.any && c->wf_tail_walk && pp.rays < std::min<unsigned long long>((unsigned long long)r.Pp / 4u, 65536ull)) {
            // The tail: no path can start any more, so ray counts only shrink from here.  Shade walks
            // the ray lists instead of the whole pool and the grids shrink.  Per-shard bound for later iterations:
            // slots never change shard and none are re-armed once the queues are empty, so no list of a shard can
            // ever grow beyond the slots alive in it now.
            pp.tail_bound = std::max<uint32_t>(64u, (pp.bound + 63u) & ~63u);
            pp.blocks_now = (uint32_t)std::min<unsigned long long>(r.trace_blocks, std::max<unsigned long long>(64, pp.rays / 32u + 64u));
        }
        // once few paths are left altogether, everything alive goes to the side pools and the pool is done
        if (!r.work_left && flush_at > 0) {
            bool ready = true;
            for (int q = 0; q < K; q++) {
                if (r.pipes[q].done) continue;
                ready = ready && r.pipes[q].any && r.pipes[q].rays <= flush_at;
            }
            if (ready) {
                uint32_t mask = 0;
                for (const WfBatch &b : r.open) mask |= 1u << b.id;
                for (int q = 0; q < K; q++) {
                    WfPipe &pq = r.pipes[q];
                    if (pq.done) continue;
                    pq.evict_next = mask;
                    rc = wf_enqueue(c, q, 1);                     // one more iteration: its shade launch empties the pool
                    if (rc) return rc;
                    pq.done = true; pq.rays = 0;
                }
                active = 0;
                r.all_evicting = true;
            }
        }
    }
    // everything enqueued for the pipes comes before the finish / resolve passes on the context's stream
    for (int p = 0; p < K; p++) {
        HIPCHK(c, hipEventRecord(c->ev_join[p], r.pipes[p].stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[p], 0));
    }
    while (!r.open.empty()) {                                    // in order: the accumulator is summed in sample order
        const WfBatch &b = r.open.front();
        if (r.all_evicting || b.evicting) {
            // (nothing else is running: few paths per wave end sooner)
            WfFinishSegs G{};
            for (int p = 0; p < K; p++) { G.ctl[G.n] = r.pipes[p].W.ctl; G.base[G.n] = r.pipes[p].W.side_base[b.id]; G.batch[G.n] = b.id; G.n++; }
            WfParams F = r.pipes[0].W;
            F.tail_bound = c->wf_flush_ppw;
            // (a batch whose eviction began before the final one may hold up to a side pool's worth)
            HIPCHK(c, wf_launch_finish(F, G, (uint32_t)std::max<unsigned long long>(std::max<unsigned long long>(flush_at, 1), b.evicting ? b.evict_bound : 0u), c->stream));
            c->last_launches++;
        }
        int rc = wf_resolve_batch(c, b);
        if (rc) return rc;
        r.open.erase(r.open.begin());
    }
    // (the status records still pending describe an empty pool; the capacity guards are checked below)
    for (int p = 0; p < K; p++) r.pipes[p].it_confirmed = r.pipes[p].it;
#ifdef CRT_WF_PROBE
    if (!c->counting) {
        // probe build: the shade kernel's phase clocks (counters 8..14 of every pipe) -> crt_debug_probes
        HIPCHK(c, hipStreamSynchronize(c->stream));
        unsigned long long pc[CRT_NCOUNTERS_DEV];
        for (int p = 0; p < K; p++) {
            {
                unsigned long long sh[kWfShards][CRT_NCOUNTERS_DEV];
                HIPCHK(c, hipMemcpy(sh, &c->w_ctl[p].p->counters[0][0], sizeof sh, hipMemcpyDeviceToHost));
                for (int k = 0; k < CRT_NCOUNTERS_DEV; k++) { pc[k] = 0; for (uint32_t s_ = 0; s_ < kWfShards; s_++) pc[k] += sh[s_][k]; }
            }
            for (int k = 0; k < 8; k++) c->probes[k] += pc[8 + k];
            HIPCHK(c, hipMemset(&c->w_ctl[p].p->counters[0][0], 0, sizeof(unsigned long long) * CRT_NCOUNTERS_DEV * kWfShards));
        }
    }
#endif
    if (c->counting) {
        // fold the pipes' counters into the context's
        HIPCHK(c, hipStreamSynchronize(c->stream));
        unsigned long long tot[CRT_NCOUNTERS];
        HIPCHK(c, hipMemcpy(tot, c->d_counters.p, sizeof tot, hipMemcpyDeviceToHost));
        unsigned long long pc[CRT_NCOUNTERS_DEV];
        for (int p = 0; p < K; p++) {
            {
                unsigned long long sh[kWfShards][CRT_NCOUNTERS_DEV];
                HIPCHK(c, hipMemcpy(sh, &c->w_ctl[p].p->counters[0][0], sizeof sh, hipMemcpyDeviceToHost));
                for (int k = 0; k < CRT_NCOUNTERS_DEV; k++) { pc[k] = 0; for (uint32_t s_ = 0; s_ < kWfShards; s_++) pc[k] += sh[s_][k]; }
            }
            for (int k = 0; k < CRT_NCOUNTERS; k++) tot[k] += pc[k];
            for (int k = 0; k < 8; k++) c->probes[k] += pc[8 + k];
            HIPCHK(c, hipMemset(&c->w_ctl[p].p->counters[0][0], 0, sizeof(unsigned long long) * CRT_NCOUNTERS_DEV * kWfShards));
        }
        HIPCHK(c, hipMemcpy(c->d_counters.p, tot, sizeof tot, hipMemcpyHostToDevice));
    }
    // (checked by wf_check_dropped after the caller's stream synchronisation: k_wf_finish's guard)
    for (int p = 0; p < K; p++)
        HIPCHK(c, hipMemcpyAsync(&c->h_dropped[p], &c->w_ctl[p].p->dropped, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    // the pool is empty; the next batch sets the pipes up afresh
    r.live = false;
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
            // (the device counter is not reset by the pool's set-up -- a flush in the middle of a run must not lose what an
            // earlier one counted -- but here, once reported)
            if (c->w_ctl[p].p) (void)hipMemsetAsync(&c->w_ctl[p].p->dropped, 0, sizeof(uint32_t), c->stream);
            return fail(c, CRT_EDEVICE, "wavefront pipeline: a capacity guard dropped %u paths (pipe %d); the frame is incomplete", n, p);
        }
    return CRT_OK;
}

int wf_trace_batch(crt_ctx *c, uint32_t n);

// Samples per batch at most: the staging buffer stays below ~6 GB and work ids fit 32 bits.
uint32_t wf_batch_cap(crt_ctx *c)
{
    const size_t npix = std::max<size_t>((size_t)c->tw * c->th, 1);
    uint32_t cap = (uint32_t)std::max<size_t>(1, std::min<size_t>(256, (size_t)6e9 / (npix * 16)));
    if (c->spp_per_launch) cap = std::min(cap, c->spp_per_launch);
    return cap;
}

// Turn the samples requested by crt_trace into batches.  Small calls are merged (option "wf_cohort", 16 samples):
// the shards of a batch's work queue are its samples, which sweep the frame together, so the paths in flight at any
// time all start inside one band of the image, many samples deep -- and the rays of a launch touch a slice of the
// scene instead of all of it (a batch of one sample per pixel has four whole frames in flight and
// costs 1.5x as much per sample).  force: publish whatever is pending (crt_sync and every call that reads state).
int wf_publish_pending(crt_ctx *c, bool force)
{
    if (c->in_publish) return CRT_OK;
    const bool defer = c->wf_defer && !c->counting;
    const uint32_t cap = wf_batch_cap(c);
    // A cohort is wf_cohort samples of a 2-Mpixel frame's worth of paths; a smaller tile (the row-band share of a
    // multi-GPU run) takes proportionally more samples, up to 8 times (measured on the 1/8 share of the 1080p frame,
    // 64 spp per call: 8.48 ms per call with every call its own batch, 7.94 ms with two calls per batch).
    const size_t npix = std::max<size_t>((size_t)c->tw * c->th, 1);
    const uint32_t scale = (uint32_t)std::min<size_t>(8, std::max<size_t>(1, ((size_t)1 << 21) / npix));
    const uint32_t cohort = c->wf_cohort <= 1 ? 1u : std::min<uint32_t>(cap, (uint32_t)c->wf_cohort * scale);
    c->in_publish = true;
    int rc = CRT_OK;
    while (c->pending > 0 && rc == CRT_OK) {
        const uint32_t take = std::min(c->pending, cap);
        if (take < cap && !force && defer && take < cohort) break;      // wait for more calls
        c->pending -= take;
        const uint32_t published0 = c->published;
        rc = wf_trace_batch(c, take);
        if (rc != CRT_OK) {                                       // what could not be published never happened
            c->sample -= take + c->pending;
            c->pending = 0;
            c->published = published0;                            // (a failure behind `published += n` drained the pool: those samples are gone too)
        }
    }
    c->in_publish = false;
    return rc;
}

// Finish whatever the pipeline still holds (no-op when nothing is in flight).
int wf_flush(crt_ctx *c)
{
    if (c->pending || (c->run && c->run->live)) HIPCHK(c, hipSetDevice(c->device));   // (publishing allocates and launches)
    if (c->pending && !c->in_publish && c->pipeline == 1 && c->accel_mode == CRT_ACCEL_BVH2) {
        int rc = wf_publish_pending(c, true);
        if (rc) return rc;
    }
    if (!c->run || !c->run->live) return CRT_OK;
    int rc = wf_finish_all(c);
    if (rc != CRT_OK) {
        // a failed drive leaves the pool in an unknown state: drain the streams and start afresh next time
        for (int p = 0; p < crt_ctx::kMaxPipes; p++) if (c->pipe_stream[p]) (void)hipStreamSynchronize(c->pipe_stream[p]);
        (void)hipStreamSynchronize(c->stream);
        c->run->live = false;
        c->run->open.clear();
        return rc;
    }
    if (c->last_timed) HIPCHK(c, hipEventRecord(c->ev1, c->stream));   // crt_last_trace_ms covers the stragglers too
    return rc;
}

// One batch of n samples through the wavefront pipeline.
int wf_trace_batch(crt_ctx *c, uint32_t n)
{
    if (!c->run) c->run = new WfRun();
    WfRun &r = *c->run;
    const WfConfig g = wf_config(c, n);
    if (g.npix == 0 || n == 0) { int rc = wf_flush(c); c->published += n; return rc; }
    // counting folds counters on the host after every batch; otherwise batches are pipelined across calls
    const bool defer = c->wf_defer && !c->counting;
    const size_t staging_elems = (size_t)n * g.npix;
    const uint32_t side_slots = kWfRing * (uint32_t)crt_ctx::kMaxPipes * kWfSideCap;   // side pools first, then the pool
    // A live pool is kept as it is unless this batch wants one more than twice as large or small (e.g. 64-spp calls
    // after 1-spp calls), or the staging buffers are too small.
    // The staging buffers are never reallocated under a live pool (the pipes hold their addresses, batches in flight
    // their contents): if ANY buffer of the ring is too small for this batch, or the ring would take the buffers past
    // the 32 GB budget at this batch's size, the pool is run to its end first and set up afresh below.
    if (r.live) {
        bool regrow = false;
        for (uint32_t b = 0; b < r.ring; b++) regrow = regrow || c->w_staging[b].n < staging_elems;
        if (r.ring > 4u && (double)r.ring * (double)staging_elems * 16.0 > 32.0e9) regrow = true;
        if ((unsigned long long)g.P > 2ull * r.P || 2ull * g.P < (unsigned long long)r.P || regrow) {
            int rc = wf_flush(c);
            if (rc) return rc;
        }
    }
    const uint32_t pool_slots = r.live ? r.P : g.P;
    const size_t list_elems = r.live ? (size_t)8 * r.list_cap * kWfShards * (size_t)r.K : g.list_per_pipe * (size_t)g.K;
    if (!r.live) {
        // batches in flight at most: the staging buffers (one per batch id) stay within ~32 GB
        uint32_t ring = std::max(2u, std::min<uint32_t>(c->wf_ring, kWfRing));
        while (ring > 4u && (double)ring * (double)staging_elems * 16.0 > 32.0e9) ring--;
        r.ring = ring;
    }
    int rc = wf_ensure(c, (size_t)pool_slots + side_slots, staging_elems, list_elems, r.ring);
    if (rc) return rc;
    if (!r.live) {
        r.K = g.K; r.P = g.P; r.Pp = g.Pp; r.list_cap = g.list_cap;
        r.trace_blocks = (uint32_t)c->num_cu * wf_waves(c);
        r.open.clear();
        WfBatch nb;
        nb.n = n; nb.last_sample = c->published + n; nb.id = 0;
        r.open.push_back(nb);
        r.seg_total[0] = g.work_total; r.seg_wps[0] = g.work_per_shard;
        for (uint32_t b = 0; b < kWfRing; b++) {
            r.queue_left[b] = false; r.consumed[b] = 0; r.resolved_recorded[b] = false;
            for (int p = 0; p < crt_ctx::kMaxPipes; p++) r.listed_until[b][p] = 0;
        }
        r.queue_left[0] = r.work_left = true;
        r.consumed_total = 0; r.rate_consumed = 0; r.rate_its = 0;                // (rate_its: set below, once the pipes' iteration numbers are)
        r.per_it = (double)g.Pp;                                 // an empty pool takes a slot's worth per slot
        r.all_evicting = false; r.poll_next = 0;
        for (int p = 0; p < r.K; p++) {
            const uint32_t it0 = r.pipes[p].it + 2u * (uint32_t)kStatusRing;
            r.pipes[p] = WfPipe();
            r.pipes[p].it = r.pipes[p].it_first = r.pipes[p].it_confirmed = r.pipes[p].it_done = it0;
            r.pipes[p].chunk = (uint32_t)c->wf_chunk;
            WfParams &W = r.pipes[p].W;
            W.sc = c->sc;
            W.ray_o = c->w_ray_o.p; W.ray_d = c->w_ray_d.p; W.sh_d = c->w_sh_d.p; W.beta = c->w_beta.p;
            W.radiance = c->w_radiance.p; W.nee = c->w_nee.p; W.rng = c->w_rng.p; W.misc = c->w_misc.p;
            W.hit = c->w_hit.p; W.vis = c->w_vis.p;
            W.dead = c->w_dead.p + (g.list_per_pipe / 8) * (size_t)p;
            W.rearm = 0;
            // k_wf_gen: waves per shard (each takes every gen_blocks-th chunk of 64 dead slots of its shard's list)
            W.gen_blocks = std::max(1u, std::min((uint32_t)c->wf_gen_blocks, g.list_cap / 64u));
            W.recA = c->w_recA.p + g.list_per_pipe * (size_t)p; W.recB = c->w_recB.p + g.list_per_pipe * (size_t)p;
            W.recC = c->w_recC.p + g.list_per_pipe * (size_t)p;
            for (uint32_t b = 0; b < kWfRing; b++) {
                W.staging[b] = c->w_staging[b < r.ring ? b : 0].p;
                W.side_base[b] = (b * (uint32_t)crt_ctx::kMaxPipes + (uint32_t)p) * kWfSideCap;
                W.seg[b] = WfSeg{0, 64, 0};
                W.seg_order[b] = 0;
            }
            W.batch_id = 0; W.count_alive = 0; W.keep_pool = 0; W.evict_mask = 0; W.status_out = nullptr;
            W.ctl = c->w_ctl[p].p; W.wq = c->w_wq.p;
            W.slot_base = side_slots + g.Pp * (uint32_t)p; W.reset_wq = (p == 0) ? 1u : 0u;
            W.P = g.Pp; W.x0 = c->x0; W.y0 = c->y0; W.tw = c->tw; W.th = c->th;
            W.band = c->band; W.stride = c->stride; W.phase = c->phase;
            W.tiles_x = g.tiles_x; W.tiles_y = g.tiles_y; W.npix_padded = g.npix_padded;
            W.list_cap = g.list_cap;
            W.seg[0] = WfSeg{g.work_total, g.work_per_shard, c->published + 1};
            W.seg_n = 1;
            W.n_samples = n;
            W.accum = accum_ptr(c); W.rgba = rgba_ptr(c);
            W.tea = c->w_tea.p;
            W.count = c->counting ? 1u : 0u;
            W.overflow_lanes = (uint32_t)c->num_cu * wf_waves(c) * 64u;
            W.stack_overflow = c->w_overflow.p + (size_t)p * W.overflow_lanes * kWfOverflowLevels;
            W.trace_form = (uint32_t)c->wf_trace_form;
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
        for (int p = 0; p < r.K; p++) r.rate_its += r.pipes[p].it_confirmed;
        // The context's stream sets the pool up and forks the pipes (and, later, finishes stragglers and
        // resolves).  The pipes run on their own streams.
        HIPCHK(c, wf_launch_init(r.pipes[0].W, c->stream));
        HIPCHK(c, wf_launch_tea(r.pipes[0].W, c->w_tea.p, c->stream));          // per-pixel RNG seed words of this tile
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        for (int p = 0; p < r.K; p++) {
            HIPCHK(c, hipStreamWaitEvent(r.pipes[p].stream, c->ev_fork, 0));
            if (p > 0) HIPCHK(c, wf_launch_init(r.pipes[p].W, r.pipes[p].stream));
        }
        // The publishing stream too: the finish / resolve passes of the PREVIOUS run may still be queued on the context's
        // stream (a flush returns without waiting for them), and the queue / side-counter reset of this run's second batch
        // must not overtake them -- resolved_recorded, which orders that reset within a run, starts afresh here.  (A reset
        // that did: k_wf_finish found side_count 0 and the batch lost its last paths -- tests: display state machine walk.)
        HIPCHK(c, hipStreamWaitEvent(c->pub_stream, c->ev_fork, 0));
        r.live = true;
    } else {
        // room in the ring first (back-pressure: the oldest batch has to retire; the pool is fed meanwhile)
        rc = wf_pump(c, true);
        if (rc) return rc;
        // The batches in flight keep their slots, queues and staging buffers; this one takes the next id and its
        // work flows into the slots that are free once the older queues are dry.
        WfBatch nb;
        nb.n = n; nb.last_sample = c->published + n; nb.id = (r.open.back().id + 1u) % r.ring;
        const uint32_t id = nb.id;
        r.seg_total[id] = g.work_total; r.seg_wps[id] = g.work_per_shard;
        r.queue_left[id] = r.work_left = true;
        r.consumed[id] = 0;
        for (int p = 0; p < r.K; p++) {
            WfPipe &pp = r.pipes[p];
            nb.from_it[p] = 0xFFFFFFFFu;                         // (set when the host sees the queue reset complete: wf_check_ready)
            pp.W.seg[id] = WfSeg{g.work_total, g.work_per_shard, c->published + 1};
            pp.W.n_samples = n;
            pp.W.batch_id = id; pp.W.keep_pool = 1;
            pp.tail_bound = 0; pp.blocks_now = r.trace_blocks;
            pp.any = false; pp.chunk = (uint32_t)c->wf_chunk; pp.done = false;
            pp.dry[id] = false; pp.alive_valid[id] = false; pp.alive[id] = 0;
        }
        // This batch's queue and side counters are reset on a stream of their own, which waits only for what it must:
        // a launch still in flight that has the id's OLD queue in its list (enqueued while that held work; it would
        // take the new work with the old batch's parameters), and the finish / resolve passes of the batch that
        // used the id before (they read its side pools and staging buffer).  No pipe waits for the reset: the queue is
        // listed by the launches that are enqueued after the host has seen the reset complete.
        for (int p = 0; p < r.K; p++)
            if (r.listed_until[id][p] > r.pipes[p].it_confirmed) {
                HIPCHK(c, hipEventRecord(c->ev_pub_join[p], r.pipes[p].stream));
                HIPCHK(c, hipStreamWaitEvent(c->pub_stream, c->ev_pub_join[p], 0));
            }
        if (r.resolved_recorded[id]) HIPCHK(c, hipStreamWaitEvent(c->pub_stream, c->ev_resolved[id], 0));
        for (int p = 0; p < r.K; p++) HIPCHK(c, wf_launch_init(r.pipes[p].W, c->pub_stream));   // (one block each)
        HIPCHK(c, hipEventRecord(c->ev_pub[id], c->pub_stream));
        nb.ready = false;                                        // listed by the launches enqueued once the host has seen that event complete
        r.open.push_back(nb);
    }
    c->published += n;
    rc = wf_pump(c, false);
    if (rc == CRT_OK && !defer) rc = wf_flush(c);
    if (rc != CRT_OK && r.live) {
        for (int p = 0; p < crt_ctx::kMaxPipes; p++) if (c->pipe_stream[p]) (void)hipStreamSynchronize(c->pipe_stream[p]);
        (void)hipStreamSynchronize(c->stream);
        r.live = false; r.open.clear();
    }
    return rc;
}

}  // namespace

extern "C" void crt_comm_on_destroy(crt_ctx *c);                 // crt_comm.cpp: the context's communicator goes with it

extern "C" {

int crt_abi_version(void) { return CRT_ABI_VERSION; }

// (for crt_comm.cpp, which is written against the public ABI: an error with the context's message)
int crt_internal_fail(crt_ctx *c, int code, const char *msg) { return fail(c, code, "%s", msg); }

int crt_get_device(crt_ctx *c, int *out)
{
    if (!c || !out) return CRT_EINVAL;
    *out = c->device;
    return CRT_OK;
}

int crt_get_stream(crt_ctx *c, void **out)
{
    if (!c || !out) return CRT_EINVAL;
    *out = (void *)c->stream;
    return CRT_OK;
}

int crt_image_size(crt_ctx *c, uint32_t out[2])
{
    if (!c || !out) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_image_size: upload a scene first");
    out[0] = c->W; out[1] = c->H;
    return CRT_OK;
}

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
    crt_comm_on_destroy(c);
    (void)hipSetDevice(c->device);
    // parked work is abandoned, but every stream must have drained before the buffers go
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int p = 0; p < crt_ctx::kMaxPipes; p++) if (c->pipe_stream[p]) (void)hipStreamSynchronize(c->pipe_stream[p]);

    delete c->run;
    c->d_raw.release(); c->d_prim.release(); c->d_primD.release(); c->d_nodes.release(); c->d_nodes4.release(); c->d_nodes4q.release(); c->d_nodes8q.release(); c->d_lights.release(); c->w_overflow.release();
    c->d_slot_of_index.release(); c->d_spectra.release(); c->d_cie.release();
    c->d_accum.release(); c->d_rgba.release(); c->d_frames.release(); c->d_counters.release();
    c->w_ray_o.release(); c->w_ray_d.release(); c->w_sh_d.release(); c->w_beta.release(); c->w_radiance.release();
    c->w_nee.release(); for (uint32_t b = 0; b < kWfRing; b++) c->w_staging[b].release(); c->w_rng.release(); c->w_misc.release(); c->w_hit.release();
    c->w_vis.release(); c->w_dead.release(); c->w_recA.release(); c->w_recB.release(); c->w_recC.release(); c->w_tea.release(); c->w_wq.release();
    if (c->pub_stream) (void)hipStreamSynchronize(c->pub_stream);
    for (int f = 0; f < crt_ctx::kFinishStreams; f++) {
        if (c->fin_stream[f]) { (void)hipStreamSynchronize(c->fin_stream[f]); (void)hipStreamDestroy(c->fin_stream[f]); }
        if (c->ev_fin[f]) (void)hipEventDestroy(c->ev_fin[f]);
    }
    for (int p = 0; p < crt_ctx::kMaxPipes; p++) {
        c->w_ctl[p].release();
        if (c->h_status[p]) (void)hipHostFree(c->h_status[p]);
        for (int k = 0; k < crt_ctx::kStatusSlots; k++) {
            if (c->ev_status[p][k]) (void)hipEventDestroy(c->ev_status[p][k]);
            if (c->ev_done[p][k]) (void)hipEventDestroy(c->ev_done[p][k]);
        }
        if (c->pipe_stream[p]) (void)hipStreamDestroy(c->pipe_stream[p]);
        if (c->ev_join[p]) (void)hipEventDestroy(c->ev_join[p]);
        if (c->ev_pub_join[p]) (void)hipEventDestroy(c->ev_pub_join[p]);
        for (uint32_t b = 0; b < kWfRing; b++) if (c->ev_evict[p][b]) (void)hipEventDestroy(c->ev_evict[p][b]);
    }
    for (uint32_t b = 0; b < kWfRing; b++) {
        if (c->ev_resolved[b]) (void)hipEventDestroy(c->ev_resolved[b]);
        if (c->ev_pub[b]) (void)hipEventDestroy(c->ev_pub[b]);
    }
    if (c->pub_stream) (void)hipStreamDestroy(c->pub_stream);
    if (c->read_stream) { (void)hipStreamSynchronize(c->read_stream); (void)hipStreamDestroy(c->read_stream); }
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
    S.nf_last[0] = S.nf_last[1] = kNoHit;
    for (size_t i = c->prims.size(); i-- > 0 && S.nf_last[1] == kNoHit;)
        if (c->prims[i].category != 2u) (S.nf_last[0] == kNoHit ? S.nf_last[0] : S.nf_last[1]) = (uint32_t)i;
    camera_frame(c->camera, S.cam);

    HIPCHK(c, c->d_raw.alloc(std::max<size_t>(nprim * 80, 16)));
    if (nprim) HIPCHK(c, hipMemcpy(c->d_raw.p, primitives, nprim * 80, hipMemcpyHostToDevice));
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
    // The build releases the scene's device arrays before it allocates the new ones: until it has succeeded there is
    // no structure to trace against (upload_geometry / build_accel_on_device set accel_mode on success only).
    c->accel_mode = -1;
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
        c->sample += n_samples; c->pending += n_samples;
        int rc = wf_tick(c);
        if (rc == CRT_OK) rc = wf_publish_pending(c, false);
        if (rc) return rc;                                       // (what was not published is not part of the frame)
    } else {
        { int rc_ = wf_flush(c); if (rc_) return rc_; }
        uint32_t chunk = c->spp_per_launch ? c->spp_per_launch : 8u;
        while (left) {
            uint32_t n = std::min(left, chunk);
            P.first_sample = c->published + 1;                            // UpdateVariables.wgsl: sample++ first
            P.n_samples = n;
            HIPCHK(c, launch_trace(P, c->counting, c->accel_mode == CRT_ACCEL_NONE, c->stream));
            c->published += n; c->sample += n; c->resolved_upto = c->published;
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

// Drive the pipeline until the resolve pass of the batch that holds `sample` is on the context's stream.
static int wf_wait_sample(crt_ctx *c, uint32_t sample)
{
    if (sample <= c->resolved_upto) return CRT_OK;
    if (c->pipeline != 1 || c->accel_mode != CRT_ACCEL_BVH2) return wf_flush(c);
    if (sample > c->published) { int rc = wf_publish_pending(c, true); if (rc) return rc; }   // (merged small calls wait for more: not any longer)
    const double t_start = wf_now_ms();
    for (int guard = 0; sample > c->resolved_upto; guard++) {
        WfRun *r = c->run;
        if (!r || !r->live || r->open.empty()) break;
        // the newest batch is only retired by a flush (nothing comes behind it under which its tail could finish)
        if (r->open.back().last_sample - r->open.back().n < sample) return wf_flush(c);
        if ((guard & 15) == 15 && wf_now_ms() - t_start > kWfStallMs) return fail(c, CRT_EDEVICE, "wavefront driver: waiting for sample %u stalled (%s)", sample, wf_state(c).c_str());
        int rc = wf_poll_all(c);
        if (rc) return rc;
        if (sample <= c->resolved_upto) break;
        rc = wf_wait_progress(c);
        if (rc) return rc;
    }
    return CRT_OK;
}

// Page-lock caller memory so that readbacks into it run at PCIe speed (an 8 MB 1080p frame: 0.15 ms instead of 1.5-2.5 ms
// through the runtime's staging of pageable memory) -- what a display loop that shows every frame wants for its frame buffer.
int crt_pin_host(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return CRT_EINVAL;
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, CRT_EDEVICE, "crt_pin_host: %s", hipGetErrorString(e)); }
    return CRT_OK;
}

int crt_unpin_host(void *ptr)
{
    if (!ptr) return CRT_EINVAL;
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, CRT_EDEVICE, "crt_unpin_host: %s", hipGetErrorString(e)); }
    return CRT_OK;
}

int crt_latest_sample(crt_ctx *c, uint32_t *out)
{
    if (!c || !out) return CRT_EINVAL;
    if (c->run && c->run->live) { HIPCHK(c, hipSetDevice(c->device)); int rc = wf_tick(c); if (rc) return rc; }   // (retire what has finished meanwhile)
    *out = c->resolved_upto;
    return CRT_OK;
}

int crt_read_latest_rgba8(crt_ctx *c, uint8_t *out, uint32_t *sample)
{
    if (!c || !out) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_read_latest_rgba8: no scene");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->run && c->run->live) { int rc = wf_tick(c); if (rc) return rc; }
    // No flush: in stream order the framebuffer holds the complete frame of the newest batch whose resolve pass has been
    // enqueued (crt_trace's contract for bound outputs); the copy is queued behind it.
    const uint32_t s = c->resolved_upto;
    const size_t n = (size_t)c->tw * c->th;
    if (n) HIPCHK(c, hipMemcpyAsync(out, rgba_ptr(c), n * sizeof(uchar4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (sample) *sample = s;
    return CRT_OK;
}

int crt_read_sample_rgba8(crt_ctx *c, uint32_t sample, uint8_t *out)
{
    if (!c || !out) return CRT_EINVAL;
    if (!c->have_scene) return fail(c, CRT_ESTATE, "crt_read_sample_rgba8: no scene");
    if (!c->frame_ring || !c->d_frames.p) return fail(c, CRT_ESTATE, "crt_read_sample_rgba8: set option frame_ring first (frames kept per sample)");
    if (sample == 0 || sample > c->sample) return fail(c, CRT_EINVAL, "crt_read_sample_rgba8: sample %u has not been requested (1..%u)", sample, c->sample);
    if (sample < c->ring_from) return fail(c, CRT_EINVAL, "crt_read_sample_rgba8: sample %u was not traced by this context since its last reset / crt_write_accum (frames from %u on)", sample, c->ring_from);
    if (c->sample - sample >= c->frame_ring) return fail(c, CRT_EINVAL, "crt_read_sample_rgba8: sample %u has left the ring of %u frames (%u requested)", sample, c->frame_ring, c->sample);
    if (c->pipeline != 1 || c->accel_mode != CRT_ACCEL_BVH2) return fail(c, CRT_ESTATE, "crt_read_sample_rgba8: frames are kept by the wavefront pipeline only");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = wf_wait_sample(c, sample); if (rc) return rc; }
    // The copy waits for the resolve pass that wrote this frame (the batch id's event: a later re-recording of it only
    // orders more), on a stream of its own -- not for the finish / resolve work of later batches queued on the context's.
    const size_t n = (size_t)c->tw * c->th;
    const uint32_t slot = (sample - 1u) % c->frame_ring;
    if (c->run && c->run->resolved_recorded[c->frame_batch[slot]]) HIPCHK(c, hipStreamWaitEvent(c->read_stream, c->ev_resolved[c->frame_batch[slot]], 0));
    else HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n) HIPCHK(c, hipMemcpyAsync(out, c->d_frames.p + (size_t)slot * n, n * sizeof(uchar4), hipMemcpyDeviceToHost, c->read_stream));
    HIPCHK(c, hipStreamSynchronize(c->read_stream));
    return CRT_OK;
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
    c->sample = sample; c->published = sample; c->pending = 0; c->resolved_upto = sample; c->ring_from = sample + 1u;
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
    if (!std::strcmp(name, "wf_ring")) { c->wf_ring = (int)std::min<int64_t>((int64_t)kWfRing, std::max<int64_t>(2, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_chunk")) { c->wf_chunk = (int)std::min<int64_t>(16, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_cohort")) { c->wf_cohort = (int)std::min<int64_t>(256, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_ahead")) { c->wf_ahead = (int)std::min<int64_t>(32, std::max<int64_t>(2, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_pool_spp")) { c->wf_pool_spp = (int)std::min<int64_t>(64, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_feed_pct")) { c->wf_feed = (double)std::min<int64_t>(400, std::max<int64_t>(10, value)) / 100.0; return CRT_OK; }
    if (!std::strcmp(name, "wf_tail_walk")) { c->wf_tail_walk = value != 0; return CRT_OK; }
    if (!std::strcmp(name, "frame_ring")) {
        c->frame_ring = (uint32_t)std::min<int64_t>(256, std::max<int64_t>(0, value));
        c->ring_from = c->sample + 1u;                          // (a new ring starts empty)
        if (c->have_scene) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipStreamSynchronize(c->stream)); return alloc_frames(c); }
        return CRT_OK;
    }
    if (!std::strcmp(name, "wf_gen_blocks")) { c->wf_gen_blocks = (int)std::min<int64_t>(4096, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_trace_form")) { c->wf_trace_form = value == 1 ? 1 : 2; return CRT_OK; }
    if (!std::strcmp(name, "wf_pipes")) { c->wf_pipes = (int)std::min<int64_t>(crt_ctx::kMaxPipes, std::max<int64_t>(1, value)); return CRT_OK; }
    if (!std::strcmp(name, "wf_pool")) { c->wf_pool = (uint32_t)std::max<int64_t>(0, value); return CRT_OK; }
    if (!std::strcmp(name, "wf_waves_per_cu")) { c->wf_waves_per_cu = (uint32_t)std::min<int64_t>(32, std::max<int64_t>(0, value)); return CRT_OK; }
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
