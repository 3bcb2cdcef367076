// crt_kernels.hip -- gfx950 path-trace kernels (the device half of libcrt.so).
//
// What runs here is the reference's per-pixel compute pass, ComputeShader.wgsl
// `main` (:77-117) with its callees, re-designed for CDNA4:
//   * one 64-lane wavefront = one 8x8 pixel tile (the WGSL workgroup), tiles
//     dealt to XCDs in contiguous image bands so each XCD's L2 keeps the part
//     of the BVH its band looks at;
//   * the reference's O(N) loop over primitives (:503-518) is replaced by a
//     BVH2 walk that returns the same hit (closest t, equal t -> later
//     primitive), with the per-lane traversal stack in LDS ([level][lane]);
//   * ONE traversal site serves both ray kinds: lanes tracing a bounce ray and
//     lanes tracing a shadow ray share the walk (shadow lanes run it any-hit
//     against the light's own t), which keeps the wave converged;
//   * `sample++` (UpdateVariables.wgsl) is folded into a kernel argument and
//     n samples are fused per launch, summed into the accumulator in sample
//     order (same f32 sum as n dispatches).
// Arithmetic follows crt_math.h to the operation so results are bit-identical
// to the CPU restatement used by the tests.
#include "crt_device.h"
#include "crt_math.h"
#include "../../include/crt.h"

namespace crt {

// ComputeShader.wgsl:11-20
#define CRT_PI 3.14159265359f
#define CRT_INFINITY 2139095040.0f   // f32(0x7F800000 as an INTEGER), :12
constexpr uint32_t kMaxDepthPath = 100;
constexpr uint32_t kGrid = 16;
constexpr uint32_t kDiffuse = 0, kLight = 1, kGlass = 2;

// ---------------------------------------------------------------- RNG (:865-897)
struct Rng { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t tea(uint32_t v0, uint32_t v1)
{
    uint32_t s0 = 0;
#pragma unroll
    for (int n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

__device__ __forceinline__ float rnd(Rng &s)
{
    s.x = s.x * 1664525u + 1013904223u;
    s.y = s.y * 1664525u + 1013904223u;
    s.z = s.z * 1664525u + 1013904223u;
    s.w = s.w * 1664525u + 1013904223u;
    s.x += s.y * s.w; s.y += s.z * s.x; s.z += s.x * s.y; s.w += s.y * s.z;
    s.x ^= s.x >> 16; s.y ^= s.y >> 16; s.z ^= s.z >> 16; s.w ^= s.w >> 16;
    s.x += s.y * s.w; s.y += s.z * s.x; s.z += s.x * s.y; s.w += s.y * s.z;
    return (float)(s.x & 0x00ffffffu) * 5.9604644775390625e-08f;   // exact: / 2^24
}

// ---------------------------------------------------------------- primitives
__device__ __forceinline__ f3 xyz(float4 v) { return f3{v.x, v.y, v.z}; }

// `t` beats the current best?  LITERAL = the reference's own test inside its
// in-order loop (:557,:609): reject only t<t_min or t>t_max, so an equal t
// from a later primitive overwrites.  Otherwise the order-independent form of
// the same rule (ties go to the larger index; NaN never wins).
template <bool LITERAL>
__device__ __forceinline__ bool beats(float t, float t_min, float t_max, uint32_t index,
                                      uint32_t b_index, uint32_t b_slot)
{
    if (LITERAL) return !(t < t_min || t > t_max);
    return (t >= t_min) && (t < t_max || (t == t_max && (b_slot == kNoHit || index > b_index)));
}

// One primitive against one ray (ComputeShader.wgsl:520-632 + category 2).
// Updates (t_max, b_index, b_slot) when the primitive becomes the best hit.
template <bool LITERAL>
__device__ __forceinline__ bool hit_test(const DevScene &S, uint32_t slot, f3 o, f3 d, uint32_t exclude,
                                         float t_min, float &t_max, uint32_t &b_index, uint32_t &b_slot)
{
    const float4 A = S.prim[3 * slot + 0];
    const float4 B = S.prim[3 * slot + 1];
    const float4 C = S.prim[3 * slot + 2];
    const uint32_t index = f_bits(B.w);
    if (exclude == index) return false;                       // :527-532
    const uint32_t cat = f_bits(A.w) & 3u;
    float t;
    if (cat == 2u) {
        // triangle: v0 = A, e1 = B, e2 = C (this project's category)
        f3 v0 = xyz(A), e1 = xyz(B), e2 = xyz(C);
        f3 pvec = cross(d, e2);
        float det = dot(e1, pvec);
        if (det == 0.0f) return false;
        float inv = 1.0f / det;
        f3 tvec = o - v0;
        float u = dot(tvec, pvec) * inv;
        if (!(u >= 0.0f && u <= 1.0f)) return false;
        f3 qvec = cross(tvec, e1);
        float v = dot(d, qvec) * inv;
        if (!(v >= 0.0f && (u + v) <= 1.0f)) return false;
        t = dot(e2, qvec) * inv;
        if (LITERAL) { if (!(t >= t_min && t <= t_max)) return false; }
        else if (!beats<false>(t, t_min, t_max, index, b_index, b_slot)) return false;
        f3 p = ray_at(o, d, t);
        f3 v1 = v0 + e1, v2 = v0 + e2;
        float pad = S.hit_pad;
        bool in = p.x >= min_(v0.x, min_(v1.x, v2.x)) - pad && p.x <= max_(v0.x, max_(v1.x, v2.x)) + pad &&
                  p.y >= min_(v0.y, min_(v1.y, v2.y)) - pad && p.y <= max_(v0.y, max_(v1.y, v2.y)) + pad &&
                  p.z >= min_(v0.z, min_(v1.z, v2.z)) - pad && p.z <= max_(v0.z, max_(v1.z, v2.z)) + pad;
        if (!in) return false;
    } else if (cat == 0u) {
        // planar patch :525-583 (unit normal and e.e precomputed with the same ops)
        const float4 D = S.primD[slot];
        f3 n = xyz(D);
        float ndotd = dot(n, d);
        if (ndotd > 0.0f) { n = -n; ndotd = -ndotd; }          // :541-545 (dot(-n,d) == -dot(n,d) exactly)
        if (abs_(ndotd) < 0.0001f) return false;               // :546
        f3 P0 = xyz(A);
        t = dot(n, P0 - o) / ndotd;                            // :554
        if (!beats<LITERAL>(t, t_min, t_max, index, b_index, b_slot)) return false;
        f3 m = ray_at(o, d, t) - P0;
        float u = dot(m, xyz(B)) / D.w;                        // :563
        float v = dot(m, xyz(C)) / C.w;                        // :564
        if (u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f) return false;
    } else {
        // sphere :584-631   A = centre, B = (r, r*r, -, index)
        f3 co = o - xyz(A);
        float a = dot(d, d);
        float b = 2.0f * dot(d, co);
        float c = dot(co, co) - B.y;
        float disc = b * b - 4.0f * a * c;
        if (disc <= 0.0f) return false;
        float sq = sqrt_(disc);
        t = (-b - sq) / (2.0f * a);
        if (LITERAL) {
            if (t < t_min || t > t_max) {
                t = (-b + sq) / (2.0f * a);
                if (t < t_min || t > t_max) return false;
            }
        } else {
            if (t < t_min) t = (-b + sq) / (2.0f * a);
            if (!beats<false>(t, t_min, t_max, index, b_index, b_slot)) return false;
        }
    }
    t_max = t; b_index = index; b_slot = slot;
    return true;
}

// Hit attributes for the winning primitive (position, shading normal).
__device__ __forceinline__ void hit_attributes(const DevScene &S, uint32_t slot, f3 o, f3 d, float t,
                                               f3 &pos, f3 &nrm, uint32_t &meta)
{
    const float4 A = S.prim[3 * slot + 0];
    meta = f_bits(A.w);
    const uint32_t cat = meta & 3u;
    pos = ray_at(o, d, t);
    if (cat == 1u) {
        nrm = normalize(pos - xyz(A));                         // :618 (always outward)
    } else {
        f3 n;
        if (cat == 0u) n = xyz(S.primD[slot]);
        else n = normalize(cross(xyz(S.prim[3 * slot + 1]), xyz(S.prim[3 * slot + 2])));
        nrm = (dot(n, d) > 0.0f) ? -n : n;                     // :541-544
    }
}

// ---------------------------------------------------------------- traversal
// The reference loop itself (:503-518): every primitive, original order.
__device__ __noinline__ void intersect_all(const DevScene &S, f3 o, f3 d, uint32_t exclude, float &t_max,
                                           uint32_t &b_index, uint32_t &b_slot, uint32_t &c_prims)
{
    for (uint32_t i = 0; i < S.nprim; i++)
        hit_test<true>(S, S.slot_of_index[i], o, d, exclude, 0.001f, t_max, b_index, b_slot);
    c_prims += S.nprim;
}

__device__ __forceinline__ bool finite3(f3 v)
{
    return abs_(v.x) < 3.0e38f && abs_(v.y) < 3.0e38f && abs_(v.z) < 3.0e38f;
}

// BVH2 walk.  stk = this lane's column of the LDS stack (stride 64 ints).
// anyhit: stop at the first primitive that beats the incoming (t_max,b_index).
template <bool COUNT>
__device__ __forceinline__ void traverse(const DevScene &S, int *stk, f3 o, f3 d, uint32_t exclude,
                                         bool anyhit, float &t_max, uint32_t &b_index, uint32_t &b_slot,
                                         uint32_t &c_nodes, uint32_t &c_prims)
{
    const float t_min = 0.001f;
    // Box culling only has to be conservative (boxes are padded); it never
    // decides a hit, so it may use any arithmetic.
    const float tiny = 1.0e-20f;
    f3 id;
    id.x = 1.0f / (abs_(d.x) > tiny ? d.x : __builtin_copysignf(tiny, d.x));
    id.y = 1.0f / (abs_(d.y) > tiny ? d.y : __builtin_copysignf(tiny, d.y));
    id.z = 1.0f / (abs_(d.z) > tiny ? d.z : __builtin_copysignf(tiny, d.z));
    const f3 oid = f3{o.x * id.x, o.y * id.y, o.z * id.z};
    const uint32_t b_slot_in = b_slot;
    int sp = 0;
    int node = S.root;
    for (;;) {
        if (node >= 0) {
            const float4 *np = S.nodes + 4 * (size_t)node;
            const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            // child 0: lo (n0.x n0.y n0.z) hi (n0.w n1.x n1.y); child 1: lo (n1.z n1.w n2.x) hi (n2.y n2.z n2.w)
            float ax = fma_(n0.x, id.x, -oid.x), bx = fma_(n0.w, id.x, -oid.x);
            float ay = fma_(n0.y, id.y, -oid.y), by = fma_(n1.x, id.y, -oid.y);
            float az = fma_(n0.z, id.z, -oid.z), bz = fma_(n1.y, id.z, -oid.z);
            float tn0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)),
                                        __builtin_fmaxf(__builtin_fminf(az, bz), t_min));
            float tf0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)),
                                        __builtin_fminf(__builtin_fmaxf(az, bz), t_max));
            ax = fma_(n1.z, id.x, -oid.x); bx = fma_(n2.y, id.x, -oid.x);
            ay = fma_(n1.w, id.y, -oid.y); by = fma_(n2.z, id.y, -oid.y);
            az = fma_(n2.x, id.z, -oid.z); bz = fma_(n2.w, id.z, -oid.z);
            float tn1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)),
                                        __builtin_fmaxf(__builtin_fminf(az, bz), t_min));
            float tf1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)),
                                        __builtin_fminf(__builtin_fmaxf(az, bz), t_max));
            if (COUNT) c_nodes += 2;
            const bool h0 = tn0 <= tf0 * 1.0000005f;
            const bool h1 = tn1 <= tf1 * 1.0000005f;
            const int r0 = (int)f_bits(n3.x), r1 = (int)f_bits(n3.y);
            if (h0 && h1) {
                const bool first0 = tn0 <= tn1;
                stk[sp * 64] = first0 ? r1 : r0;
                sp++;
                node = first0 ? r0 : r1;
                continue;
            }
            if (h0) { node = r0; continue; }
            if (h1) { node = r1; continue; }
        } else {
            const uint32_t enc = ~(uint32_t)node;
            const uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++)
                hit_test<false>(S, first + i, o, d, exclude, t_min, t_max, b_index, b_slot);
            if (COUNT) c_prims += cnt;
            if (anyhit && b_slot != b_slot_in) return;
        }
        if (sp == 0) return;
        sp--;
        node = stk[sp * 64];
    }
}

// ---------------------------------------------------------------- shading helpers
__device__ __forceinline__ f4 sample_spectrum(const DevScene &S, uint32_t index, const uint32_t l[4])
{
    if (index >= S.nspectra) index = S.nspectra - 1u;          // robust-access clamp (Q7 pin)
    const float *row = S.spectra + (size_t)index * kNLambda;
    return f4{row[l[0]], row[l[1]], row[l[2]], row[l[3]]};
}

__device__ __forceinline__ float power_heuristic(float nf, float f_pdf, float ng, float g_pdf)
{
    float f = nf * f_pdf, g = ng * g_pdf;                      // :297-302
    return (f * f) / (f * f + g * g);
}

// :357-377.  `emission_index` indexes lights[] (sic, Q7), clamped.
__device__ __forceinline__ float compute_light_pdf(const DevScene &S, uint32_t emission_index, f3 position,
                                                   f3 normal, f3 ray_origin, f3 ray_direction)
{
    uint32_t li = emission_index;
    if (li >= S.nlight) li = S.nlight - 1u;
    float light_area_pdf = S.lights[3 * li + 2].w;             // 1.0 / (|data2| * |data3|)
    float abs_cos_theta = max_(0.00001f, abs_(dot(normal, -ray_direction)));
    float distance = length(position - ray_origin);
    float distance_squared = pow_(distance, 2.0f);             // :368
    float geometric_term = abs_cos_theta / distance_squared;
    float light_solid_angle_pdf = light_area_pdf / geometric_term;
    return S.inv_nlight * light_solid_angle_pdf;
}

// :814-837
__device__ __forceinline__ float fresnel_s(f3 ray_dir, f3 normal, float eta1, float eta2)
{
    float cosi = min_(max_(dot(ray_dir, normal), -1.0f), 1.0f);
    float eta = eta1 / eta2;
    if (cosi > 0.0f) eta = eta2 / eta1;
    float sint2 = eta * eta * (1.0f - cosi * cosi);
    if (sint2 > 1.0f) return 1.0f;
    float cost = sqrt_(1.0f - sint2);
    cosi = abs_(cosi);
    float Rs = ((eta1 * cosi) - (eta2 * cost)) / ((eta1 * cosi) + (eta2 * cost));
    float Rp = ((eta2 * cosi) - (eta1 * cost)) / ((eta2 * cosi) + (eta1 * cost));
    return (Rs * Rs + Rp * Rp) / 2.0f;
}

__device__ __forceinline__ f3 reflect_(f3 e1, f3 e2)
{
    float k = 2.0f * dot(e2, e1);
    return e1 - e2 * k;
}
__device__ __forceinline__ f3 refract_(f3 e1, f3 e2, float e3)
{
    float dd = dot(e2, e1);
    float k = 1.0f - e3 * e3 * (1.0f - dd * dd);
    if (k < 0.0f) return f3{0.0f, 0.0f, 0.0f};
    float s = e3 * dd + sqrt_(k);
    return e1 * e3 - e2 * s;
}

// :751-774
__device__ __forceinline__ f3 cosine_hemisphere(Rng &rng, f3 normal, float &pdf)
{
    float u = rnd(rng);
    float v = rnd(rng);
    float r = sqrt_(u);
    float theta = (2.0f * CRT_PI) * v;
    float st, ct;
    sincos_(theta, st, ct);
    float x = r * ct;
    float y = r * st;
    float z = sqrt_(max_(0.0f, 1.0f - u));
    f3 up = (abs_(normal.z) < 0.999f) ? f3{0.0f, 0.0f, 1.0f} : f3{1.0f, 0.0f, 0.0f};
    f3 tangent = normalize(cross(up, normal));
    f3 bitangent = cross(normal, tangent);
    f3 dir = (tangent * x + bitangent * y) + normal * z;
    pdf = z / CRT_PI;
    return dir;
}

__device__ __forceinline__ uint8_t unorm8(float x)
{
    if (!(x > 0.0f)) return 0;
    if (x > 1.0f) x = 1.0f;
    return (uint8_t)(x * 255.0f + 0.5f);
}

__device__ __forceinline__ float gamma_rb(float c)
{
    return (c < 0.0031308f) ? c * 12.92f : 1.055f * pow_(c, (float)(1.0 / 2.4)) - 0.055f;
}

// Wave-wide sum then one atomic per wave (Guideline 12).
__device__ __forceinline__ void wave_add(unsigned long long *dst, uint32_t v)
{
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst, s);
}

// ---------------------------------------------------------------- the kernel
// grid = tiles_x*tiles_y blocks of 64 threads; block -> 8x8 pixel tile of the
// context's rectangle.  BRUTE: no BVH, the reference loop for every ray.
template <bool COUNT, bool BRUTE>
__global__ __launch_bounds__(64) void k_trace(const TraceParams P)
{
    __shared__ int lds_stack[kStackDepth * 64];
    const DevScene &S = P.sc;
    const uint32_t lane = threadIdx.x;
    int *stk = lds_stack + lane;

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin
    // dispatch), so give each XCD a contiguous band of tiles (bijective form).
    const uint32_t ntiles = P.tiles_x * P.tiles_y;
    const uint32_t b = blockIdx.x;
    const uint32_t q = ntiles >> 3, r = ntiles & 7u, xcd = b & 7u;
    const uint32_t tile = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (b >> 3);
    const uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const uint32_t lx = tx * 8u + (lane & 7u), ly = ty * 8u + (lane >> 3);
    const bool in_tile = lx < P.tw && ly < P.th;
    const uint32_t px = P.x0 + lx, py = P.y0 + ly;             // GLOBAL pixel (seeds, film position)
    const size_t pix = (size_t)lx + (size_t)ly * P.tw;

    uint32_t c_rays = 0, c_nodes = 0, c_prims = 0, c_bounces = 0, c_shadow = 0, c_hits = 0;

    if (in_tile) {
        float4 acc4 = P.accum[pix];
        f3 acc = f3{acc4.x, acc4.y, acc4.z};
        const uint32_t tea_xy = tea(px, py * 100u);
        const f3 llc = f3{S.cam[0], S.cam[1], S.cam[2]}, hor = f3{S.cam[3], S.cam[4], S.cam[5]};
        const f3 ver = f3{S.cam[6], S.cam[7], S.cam[8]}, eye = f3{S.cam[9], S.cam[10], S.cam[11]};
        uint32_t sample = P.first_sample;

        for (uint32_t si = 0; si < P.n_samples; si++, sample++) {
            Rng rng = {py, px * 100u, sample, tea_xy};                              // :98
            // camera_ray :477-500
            float jx = rnd(rng);
            float fs = ((float)px + ((float)(sample % kGrid) + jx) / (float)kGrid) / (float)S.W;
            float jy = rnd(rng);
            float ft = ((float)S.H - (float)py + ((float)(sample % kGrid) + jy) / (float)kGrid) / (float)S.H;
            f3 ray_o = eye;
            f3 ray_d = normalize(((llc + hor * fs) + ver * ft) - eye);
            // sample_wavelengths :315-322
            uint32_t wl[4];
            {
                float u = rnd(rng);
                uint32_t lambda = (uint32_t)(301.0f * u);
                wl[0] = lambda; wl[1] = (lambda + 4u) % kNLambda; wl[2] = (lambda + 8u) % kNLambda;
                wl[3] = (lambda + 12u) % kNLambda;
            }
            // path_trace :119-295, as a two-state loop around ONE traversal site
            uint32_t depth = 0;
            f4 radiance = f4{0, 0, 0, 0};
            f4 beta = f4{1, 1, 1, 1};
            float last_bounce_pdf = 1.0f;
            uint32_t exclude = 0xFFFFFFFFu;
            bool specular_bounce = false;
            float etaScale = 1.0f;
            bool inTransmission = false;
            // pending-shadow state (valid while shadow_mode)
            bool shadow_mode = false;
            f3 s_pos = f3{0, 0, 0}, s_nrm = f3{0, 0, 0}, s_ldir = f3{0, 0, 0};
            f4 s_brdf = f4{0, 0, 0, 0};
            uint32_t s_light = 0, s_index = 0;

            for (;;) {
                // ---- the ray this lane traces now
                f3 o = shadow_mode ? s_pos : ray_o;
                f3 d = shadow_mode ? s_ldir : ray_d;
                uint32_t excl = shadow_mode ? s_index : exclude;
                float t_max = CRT_INFINITY;
                uint32_t b_index = kNoHit, b_slot = kNoHit;
                bool occluded = false, light_seen = false;
                if (COUNT) { c_rays++; if (shadow_mode) c_shadow++; }
                if (BRUTE || !finite3(o) || !finite3(d)) {
                    intersect_all(S, o, d, excl, t_max, b_index, b_slot, c_prims);
                    if (shadow_mode) {
                        uint32_t include = f_bits(S.lights[3 * s_light + 1].w);
                        light_seen = (b_slot != kNoHit && b_index == include);     // :700
                        occluded = !light_seen;
                    }
                } else if (shadow_mode) {
                    // shadow_intersect (:697-705) == "is the light's own primitive the closest
                    // hit?"  Hit the light first, then ask the BVH for anything that beats it.
                    uint32_t include = f_bits(S.lights[3 * s_light + 1].w);
                    if (include < S.nprim) {
                        hit_test<false>(S, S.slot_of_index[include], o, d, excl, 0.001f, t_max, b_index, b_slot);
                        if (COUNT) c_prims++;
                    }
                    if (b_slot != kNoHit) {
                        const uint32_t l_slot = b_slot;
                        traverse<COUNT>(S, stk, o, d, excl, true, t_max, b_index, b_slot, c_nodes, c_prims);
                        occluded = (b_slot != l_slot);
                        light_seen = !occluded;
                    } else {
                        occluded = true;
                    }
                } else {
                    traverse<COUNT>(S, stk, o, d, excl, false, t_max, b_index, b_slot, c_nodes, c_prims);
                }

                bool diffuse_tail = false;   // run the second half of the DIFFUSE block
                f4 nee = f4{0, 0, 0, 0};
                f3 h_pos = s_pos, h_nrm = s_nrm;

                if (shadow_mode) {
                    // ---- compute_light_radiance tail :388-407
                    if (light_seen) {
                        f3 lp, ln; uint32_t lmeta;
                        hit_attributes(S, b_slot, o, d, t_max, lp, ln, lmeta);
                        if (COUNT) c_hits++;
                        float cos_theta = max_(0.0f, dot(s_nrm, s_ldir));
                        f4 spec = sample_spectrum(S, f_bits(S.lights[3 * s_light + 0].w), wl);
                        f4 le = spec * cos_theta;
                        float pdf_l = compute_light_pdf(S, (lmeta >> 4) & 0x3FFFu, lp, ln, o, d);
                        float pdf_b = cos_theta / CRT_PI;
                        float weight_l = power_heuristic(1.0f, pdf_l, 1.0f, pdf_b);
                        nee = (le * weight_l) / pdf_l;
                    }
                    shadow_mode = false;
                    diffuse_tail = true;
                } else {
                    if (COUNT) c_bounces++;
                    if (b_slot == kNoHit) break;                                     // :141
                    f3 pos, nrm; uint32_t meta;
                    hit_attributes(S, b_slot, o, d, t_max, pos, nrm, meta);
                    if (COUNT) c_hits++;
                    exclude = b_index;                                               // :146
                    const uint32_t material = (meta >> 2) & 3u;
                    const uint32_t emission_index = (meta >> 4) & 0x3FFFu;
                    const uint32_t reflectance_index = (meta >> 18) & 0x3FFFu;
                    if (material == kLight) {                                        // :149-164
                        f4 le = sample_spectrum(S, emission_index, wl);
                        if (depth == 0 || specular_bounce) {
                            radiance = radiance + beta * le;
                        } else {
                            float pdf_l = compute_light_pdf(S, emission_index, pos, nrm, o, d);
                            float weight_b = power_heuristic(1.0f, last_bounce_pdf, 1.0f, pdf_l);
                            radiance = radiance + (le * weight_b) * beta;
                        }
                        break;
                    }
                    if (depth >= kMaxDepthPath) break;                               // :167
                    if (inTransmission) {                                            // :173-179
                        float distance = length(o - pos);
                        f4 ext = sample_spectrum(S, S.nspectra - 1u, wl);
                        f4 att = f4{exp_(-ext.x * distance), exp_(-ext.y * distance), exp_(-ext.z * distance),
                                    exp_(-ext.w * distance)};
                        beta = beta * att;
                    }
                    if (material == kDiffuse) {                                      // :182-204, first half
                        s_brdf = sample_spectrum(S, reflectance_index, wl) / CRT_PI;
                        // sample_lights :341-347, sample_light :349-355
                        float u0 = rnd(rng);
                        uint32_t li = (uint32_t)((float)S.nlight * u0);
                        if (li >= S.nlight) li = S.nlight - 1u;
                        float u = rnd(rng);
                        float v = rnd(rng);
                        f3 l1 = xyz(S.lights[3 * li + 0]), l2 = xyz(S.lights[3 * li + 1]), l3 = xyz(S.lights[3 * li + 2]);
                        f3 pl = (l1 + l2 * u) + l3 * v;
                        s_ldir = normalize(pl - pos);
                        s_pos = pos; s_nrm = nrm; s_light = li; s_index = b_index;
                        shadow_mode = true;
                        continue;                                                    // trace the shadow ray
                    }
                    if (material == kGlass) {                                        // :208-276
                        const float eta1 = 1.0f, eta2 = 1.5f;
                        float eta = eta1 / eta2;
                        float cos_theta = dot(nrm, ray_d);
                        float reflected = fresnel_s(ray_d, nrm, eta1, eta2);
                        float pr = reflected;
                        float pt = 1.0f - reflected;
                        float u = rnd(rng);
                        f3 current_normal = nrm;
                        if (cos_theta > 0.0f) { eta = 1.0f / eta; current_normal = -current_normal; }
                        f3 new_direction;
                        if (u < pr / (pr + pt)) {                                    // :238
                            new_direction = reflect_(ray_d, current_normal);
                        } else {
                            new_direction = normalize(refract_(ray_d, current_normal, eta));
                            beta = beta * (eta * eta);
                            etaScale = etaScale / (eta * eta);
                            inTransmission = !inTransmission;
                        }
                        ray_o = pos;
                        specular_bounce = true;
                        exclude = 0xFFFFFFFFu;
                        ray_d = new_direction;
                    }
                    h_pos = pos; h_nrm = nrm;
                }

                if (diffuse_tail) {                                                  // :187-195
                    radiance = radiance + (s_brdf * nee) * beta;
                    f3 new_direction = cosine_hemisphere(rng, h_nrm, last_bounce_pdf);
                    float cos_theta = abs_(dot(h_nrm, new_direction));
                    beta = beta * ((s_brdf * cos_theta) / last_bounce_pdf);
                    ray_o = h_pos;
                    ray_d = new_direction;
                    specular_bounce = false;
                }
                // Russian roulette :279-289
                {
                    f4 rbeta = beta * etaScale;
                    float max_beta_component = max_(rbeta.x, max_(rbeta.y, rbeta.z));
                    if (depth > 1u && max_beta_component < 1.0f) {
                        float qq = max_(0.0f, 1.0f - max_beta_component);
                        if (rnd(rng) < qq) break;
                        beta = beta / (1.0f - qq);
                    }
                }
                depth++;
            }

            // spectral_to_xyz :419-426
            const float *X = S.cie, *Y = S.cie + kNCie, *Z = S.cie + 2 * kNCie;
            f4 xb = f4{X[wl[0] + 40], X[wl[1] + 40], X[wl[2] + 40], X[wl[3] + 40]};
            f4 yb = f4{Y[wl[0] + 40], Y[wl[1] + 40], Y[wl[2] + 40], Y[wl[3] + 40]};
            f4 zb = f4{Z[wl[0] + 40], Z[wl[1] + 40], Z[wl[2] + 40], Z[wl[3] + 40]};
            f3 xyzc = f3{dot(xb, radiance), dot(yb, radiance), dot(zb, radiance)};
            xyzc = (xyzc * 300.0f) / (106.856895f * 4.0f);
            acc = acc + xyzc;                                                        // :108
        }

        P.accum[pix] = float4{acc.x, acc.y, acc.z, acc4.w};
        if (P.n_samples > 0) {
            // :110-115  average, XYZ->sRGB, exposure tone map, gamma (G-channel bug kept, Q10)
            f3 avg = acc / (float)(sample - 1u);
            float rr = 3.2404542f * avg.x + -1.5371385f * avg.y + -0.4985314f * avg.z;
            float gg = -0.9692660f * avg.x + 1.8760108f * avg.y + 0.0415560f * avg.z;
            float bb = 0.0556434f * avg.x + -0.2040259f * avg.y + 1.0572252f * avg.z;
            rr = 1.0f - exp_(-rr * 2.2f);
            gg = 1.0f - exp_(-gg * 2.2f);
            bb = 1.0f - exp_(-bb * 2.2f);
            rr = gamma_rb(rr);
            gg = (gg < 0.0031308f) ? gg * (12.92f * gg) : 1.055f * pow_(gg, (float)(1.0 / 2.4)) - 0.055f;
            bb = (bb < 0.0031308f) ? 12.92f * bb : 1.055f * pow_(bb, (float)(1.0 / 2.4)) - 0.055f;
            P.rgba[pix] = uchar4{unorm8(rr), unorm8(gg), unorm8(bb), 255};
        }
    }

    if (COUNT && P.counters) {
        wave_add(P.counters + CRT_CNT_RAYS, c_rays);
        wave_add(P.counters + CRT_CNT_NODES, c_nodes);
        wave_add(P.counters + CRT_CNT_PRIMS, c_prims);
        wave_add(P.counters + CRT_CNT_BOUNCES, c_bounces);
        wave_add(P.counters + CRT_CNT_SHADOW, c_shadow);
        wave_add(P.counters + CRT_CNT_HITS, c_hits);
        wave_add(P.counters + CRT_CNT_PATHS, in_tile ? P.n_samples : 0u);
    }
}

// ---------------------------------------------------------------- test hooks
__global__ void k_debug_intersect(const DevScene S, const float *rays, size_t n, float *out, int brute)
{
    __shared__ int lds_stack[kStackDepth * 64];
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    int *stk = lds_stack + threadIdx.x;
    f3 o = f3{rays[8 * i + 0], rays[8 * i + 1], rays[8 * i + 2]};
    f3 d = f3{rays[8 * i + 3], rays[8 * i + 4], rays[8 * i + 5]};
    uint32_t excl = f_bits(rays[8 * i + 6]);
    float t_max = CRT_INFINITY;
    uint32_t b_index = kNoHit, b_slot = kNoHit, cn = 0, cp = 0;
    if (brute || !finite3(o) || !finite3(d)) intersect_all(S, o, d, excl, t_max, b_index, b_slot, cp);
    else traverse<false>(S, stk, o, d, excl, false, t_max, b_index, b_slot, cn, cp);
    f3 pos = f3{0, 0, 0}, nrm = f3{0, 0, 0};
    uint32_t meta = 0;
    if (b_slot != kNoHit) hit_attributes(S, b_slot, o, d, t_max, pos, nrm, meta);
    float *r = out + 8 * i;
    r[0] = t_max; r[1] = pos.x; r[2] = pos.y; r[3] = pos.z; r[4] = nrm.x; r[5] = nrm.y; r[6] = nrm.z;
    r[7] = bits_f(b_slot != kNoHit ? b_index : kNoHit);
}

__global__ void k_debug_math(int fn, const float *a, const float *b, float *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], r;
    switch (fn) {
    case 0: r = sin_(x); break;
    case 1: r = cos_(x); break;
    case 2: r = exp_(x); break;
    case 3: r = log2_(x); break;
    case 4: r = exp2_(x); break;
    case 5: r = pow_(x, y); break;
    case 6: r = sqrt_(x); break;
    case 7: r = x / y; break;
    case 8: r = tan_(x); break;
    default: r = 0.0f;
    }
    out[i] = r;
}

// ---------------------------------------------------------------- launchers (called from crt_api.cpp)
hipError_t launch_trace(const TraceParams &P, bool count, bool brute, hipStream_t stream)
{
    dim3 grid(P.tiles_x * P.tiles_y), block(64);
    if (grid.x == 0) return hipSuccess;
    if (count) {
        if (brute) hipLaunchKernelGGL((k_trace<true, true>), grid, block, 0, stream, P);
        else hipLaunchKernelGGL((k_trace<true, false>), grid, block, 0, stream, P);
    } else {
        if (brute) hipLaunchKernelGGL((k_trace<false, true>), grid, block, 0, stream, P);
        else hipLaunchKernelGGL((k_trace<false, false>), grid, block, 0, stream, P);
    }
    return hipGetLastError();
}

hipError_t launch_debug_intersect(const DevScene &S, const float *rays, size_t n, float *out, int brute,
                                  hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_debug_intersect, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, S, rays, n, out, brute);
    return hipGetLastError();
}

hipError_t launch_debug_math(int fn, const float *a, const float *b, float *out, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fn, a, b, out, n);
    return hipGetLastError();
}

}  // namespace crt
