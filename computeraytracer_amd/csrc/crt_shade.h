// crt_shade.h -- device functions shared by the trace kernels: the integer RNG,
// primitive tests, hit attributes and the shading helpers of ComputeShader.wgsl,
// each cited at its definition.  Arithmetic follows crt_math.h to the operation.
#pragma once
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

// One primitive against one ray (ComputeShader.wgsl:520-632 + category 2), the primitive given as its record
// (A, B, C of `prim`, D of `primD` -- D is looked at for patches only).
// Updates (t_max, b_index, b_slot) when the primitive becomes the best hit.
template <bool LITERAL>
__device__ __forceinline__ bool hit_test_rec(const float4 A, const float4 B, const float4 C, const float4 D, float hit_pad,
                                             uint32_t slot, f3 o, f3 d, uint32_t exclude, float t_min, float &t_max,
                                             uint32_t &b_index, uint32_t &b_slot)
{
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
        float pad = hit_pad;
        bool in = p.x >= min_(v0.x, min_(v1.x, v2.x)) - pad && p.x <= max_(v0.x, max_(v1.x, v2.x)) + pad &&
                  p.y >= min_(v0.y, min_(v1.y, v2.y)) - pad && p.y <= max_(v0.y, max_(v1.y, v2.y)) + pad &&
                  p.z >= min_(v0.z, min_(v1.z, v2.z)) - pad && p.z <= max_(v0.z, max_(v1.z, v2.z)) + pad;
        if (!in) return false;
    } else if (cat == 0u) {
        // planar patch :525-583 (unit normal and e.e precomputed with the same ops)
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

// The same with the record fetched from the scene (primD only for a patch that is not excluded).
template <bool LITERAL>
__device__ __forceinline__ bool hit_test(const DevScene &S, uint32_t slot, f3 o, f3 d, uint32_t exclude,
                                         float t_min, float &t_max, uint32_t &b_index, uint32_t &b_slot)
{
    const float4 A = S.prim[3 * slot + 0];
    const float4 B = S.prim[3 * slot + 1];
    const float4 C = S.prim[3 * slot + 2];
    float4 D = float4{0.0f, 0.0f, 0.0f, 0.0f};
    if (exclude != f_bits(B.w) && (f_bits(A.w) & 3u) == 0u) D = S.primD[slot];
    return hit_test_rec<LITERAL>(A, B, C, D, S.hit_pad, slot, o, d, exclude, t_min, t_max, b_index, b_slot);
}

// Hit attributes for the winning primitive (position, shading normal), from its record.
__device__ __forceinline__ void hit_attributes_rec(const float4 A, const float4 B, const float4 C, const float4 D, f3 o, f3 d,
                                                   float t, f3 &pos, f3 &nrm, uint32_t &meta)
{
    meta = f_bits(A.w);
    const uint32_t cat = meta & 3u;
    pos = ray_at(o, d, t);
    if (cat == 1u) {
        nrm = normalize(pos - xyz(A));                         // :618 (always outward)
    } else {
        f3 n;
        if (cat == 0u) n = xyz(D);
        else n = normalize(cross(xyz(B), xyz(C)));
        nrm = (dot(n, d) > 0.0f) ? -n : n;                     // :541-544
    }
}

__device__ __forceinline__ void hit_attributes(const DevScene &S, uint32_t slot, f3 o, f3 d, float t,
                                               f3 &pos, f3 &nrm, uint32_t &meta)
{
    const float4 A = S.prim[3 * slot + 0];
    const uint32_t cat = f_bits(A.w) & 3u;
    float4 B = float4{0.0f, 0.0f, 0.0f, 0.0f}, C = B, D = B;
    if (cat == 0u) D = S.primD[slot];
    else if (cat == 2u) { B = S.prim[3 * slot + 1]; C = S.prim[3 * slot + 2]; }
    hit_attributes_rec(A, B, C, D, o, d, t, pos, nrm, meta);
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
    if (S.nprim == 0u) return;
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

// Single-ray walk of the quantised 4-wide tree (the nodes k_wf_trace walks: 64 bytes, 16-bit planes; same culling
// arithmetic): less than half the dependent node fetches of the BVH2 walk above, which is what a lane of k_wf_finish
// pays for -- a straggler's bounce is as long as its two walks.  Returns false when the LDS stack (kStackDepth
// entries) would overflow: the caller repeats the ray with traverse(), whose stack need is bounded by the builders.
template <bool COUNT>
__device__ __forceinline__ bool traverse4q(const DevScene &S, int *stk, f3 o, f3 d, uint32_t exclude, bool anyhit,
                                           float &t_max, uint32_t &b_index, uint32_t &b_slot, uint32_t &c_nodes, uint32_t &c_prims)
{
    const float t_min = 0.001f;
    if (S.nprim == 0u) return true;
    const float tiny = 1.0e-20f;
    f3 id;
    id.x = 1.0f / (abs_(d.x) > tiny ? d.x : __builtin_copysignf(tiny, d.x));
    id.y = 1.0f / (abs_(d.y) > tiny ? d.y : __builtin_copysignf(tiny, d.y));
    id.z = 1.0f / (abs_(d.z) > tiny ? d.z : __builtin_copysignf(tiny, d.z));
    const bool gx = id.x < 0.0f, gy = id.y < 0.0f, gz = id.z < 0.0f;
    // plane = qbase + q * qscale  =>  t = q * (qscale * id) + (qbase * id - o * id): one fma per plane
    const f3 oid = f3{fma_(S.qbase[0], id.x, -(o.x * id.x)), fma_(S.qbase[1], id.y, -(o.y * id.y)), fma_(S.qbase[2], id.z, -(o.z * id.z))};
    id = f3{S.qscale[0] * id.x, S.qscale[1] * id.y, S.qscale[2] * id.z};
    const uint32_t b_slot_in = b_slot;
    const float t_in = t_max;
    const uint32_t i_in = b_index;
    int sp = 0;
    int node = S.root4;
    for (;;) {
        if (node >= 0) {
            const uint4 *nq = S.nodes4q + 4 * (size_t)node;
            const uint4 Q0 = nq[0], Q1 = nq[1], Q2 = nq[2], Q3 = nq[3];
            const uint32_t nxa = gx ? Q1.z : Q0.x, nxb = gx ? Q1.w : Q0.y, fxa = gx ? Q0.x : Q1.z, fxb = gx ? Q0.y : Q1.w;
            const uint32_t nya = gy ? Q2.x : Q0.z, nyb = gy ? Q2.y : Q0.w, fya = gy ? Q0.z : Q2.x, fyb = gy ? Q0.w : Q2.y;
            const uint32_t nza = gz ? Q2.z : Q1.x, nzb = gz ? Q2.w : Q1.y, fza = gz ? Q1.x : Q2.z, fzb = gz ? Q1.y : Q2.w;
            float k0, k1, k2, k3;
#define CRT_QBOX1(K, NXQ, NYQ, NZQ, FXQ, FYQ, FZQ) { \
                const float tn_ = __builtin_fmaxf(__builtin_fmaxf(fma_((float)(NXQ), id.x, oid.x), fma_((float)(NYQ), id.y, oid.y)), \
                                                  __builtin_fmaxf(fma_((float)(NZQ), id.z, oid.z), t_min)); \
                const float tf_ = __builtin_fminf(__builtin_fminf(fma_((float)(FXQ), id.x, oid.x), fma_((float)(FYQ), id.y, oid.y)), \
                                                  __builtin_fminf(fma_((float)(FZQ), id.z, oid.z), t_max)); \
                K = (tn_ <= tf_ * 1.0000005f) ? tn_ : 3.0e38f; }
            CRT_QBOX1(k0, nxa & 0xFFFFu, nya & 0xFFFFu, nza & 0xFFFFu, fxa & 0xFFFFu, fya & 0xFFFFu, fza & 0xFFFFu)
            CRT_QBOX1(k1, nxa >> 16, nya >> 16, nza >> 16, fxa >> 16, fya >> 16, fza >> 16)
            CRT_QBOX1(k2, nxb & 0xFFFFu, nyb & 0xFFFFu, nzb & 0xFFFFu, fxb & 0xFFFFu, fyb & 0xFFFFu, fzb & 0xFFFFu)
            CRT_QBOX1(k3, nxb >> 16, nyb >> 16, nzb >> 16, fxb >> 16, fyb >> 16, fzb >> 16)
#undef CRT_QBOX1
            int r0 = (int)Q3.x, r1 = (int)Q3.y, r2 = (int)Q3.z, r3 = (int)Q3.w;
            if (COUNT) c_nodes += 4;
#define CRT_CAS1(ka, ra, kb, rb) { const bool sw_ = kb < ka; const float tk_ = sw_ ? kb : ka; kb = sw_ ? ka : kb; ka = tk_; \
                                   const int tr_ = sw_ ? rb : ra; rb = sw_ ? ra : rb; ra = tr_; }
            CRT_CAS1(k0, r0, k1, r1) CRT_CAS1(k2, r2, k3, r3) CRT_CAS1(k0, r0, k2, r2) CRT_CAS1(k1, r1, k3, r3) CRT_CAS1(k1, r1, k2, r2)
#undef CRT_CAS1
            if (k0 < 3.0e38f) {
                // descend into the nearest; the others wait on the stack, farthest pushed first
                const int np = (k1 < 3.0e38f ? 1 : 0) + (k2 < 3.0e38f ? 1 : 0) + (k3 < 3.0e38f ? 1 : 0);
                if (sp + np > kStackDepth) { t_max = t_in; b_index = i_in; b_slot = b_slot_in; return false; }
                if (k3 < 3.0e38f) { stk[sp * 64] = r3; sp++; }
                if (k2 < 3.0e38f) { stk[sp * 64] = r2; sp++; }
                if (k1 < 3.0e38f) { stk[sp * 64] = r1; sp++; }
                node = r0;
                continue;
            }
        } else {
            const uint32_t enc = ~(uint32_t)node;
            const uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++)
                hit_test<false>(S, first + i, o, d, exclude, t_min, t_max, b_index, b_slot);
            if (COUNT) c_prims += cnt;
            if (anyhit && b_slot != b_slot_in) return true;
        }
        if (sp == 0) return true;
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

// :357-377 with the light's 1/area given.
__device__ __forceinline__ float compute_light_pdf_area(float light_area_pdf, float inv_nlight, f3 position,
                                                        f3 normal, f3 ray_origin, f3 ray_direction)
{
    float abs_cos_theta = max_(0.00001f, abs_(dot(normal, -ray_direction)));
    float distance = length(position - ray_origin);
    float distance_squared = pow_(distance, 2.0f);             // :368
    float geometric_term = abs_cos_theta / distance_squared;
    float light_solid_angle_pdf = light_area_pdf / geometric_term;
    return inv_nlight * light_solid_angle_pdf;
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


// :110-115  average, XYZ->sRGB, exposure tone map, gamma (G-channel bug kept, Q10), rgba8unorm store
__device__ __forceinline__ uchar4 tonemap_rgba8(f3 acc, float n_samples_f)
{
    f3 avg = acc / n_samples_f;
    float rr = 3.2404542f * avg.x + -1.5371385f * avg.y + -0.4985314f * avg.z;
    float gg = -0.9692660f * avg.x + 1.8760108f * avg.y + 0.0415560f * avg.z;
    float bb = 0.0556434f * avg.x + -0.2040259f * avg.y + 1.0572252f * avg.z;
    rr = 1.0f - exp_(-rr * 2.2f);
    gg = 1.0f - exp_(-gg * 2.2f);
    bb = 1.0f - exp_(-bb * 2.2f);
    rr = gamma_rb(rr);
    gg = (gg < 0.0031308f) ? gg * (12.92f * gg) : 1.055f * pow_(gg, (float)(1.0 / 2.4)) - 0.055f;
    bb = (bb < 0.0031308f) ? 12.92f * bb : 1.055f * pow_(bb, (float)(1.0 / 2.4)) - 0.055f;
    return uchar4{unorm8(rr), unorm8(gg), unorm8(bb), 255};
}

// spectral_to_xyz :419-426
__device__ __forceinline__ f3 spectral_to_xyz(const DevScene &S, f4 radiance, const uint32_t wl[4])
{
    const float *X = S.cie, *Y = S.cie + kNCie, *Z = S.cie + 2 * kNCie;
    f4 xb = f4{X[wl[0] + 40], X[wl[1] + 40], X[wl[2] + 40], X[wl[3] + 40]};
    f4 yb = f4{Y[wl[0] + 40], Y[wl[1] + 40], Y[wl[2] + 40], Y[wl[3] + 40]};
    f4 zb = f4{Z[wl[0] + 40], Z[wl[1] + 40], Z[wl[2] + 40], Z[wl[3] + 40]};
    f3 xyzc = f3{dot(xb, radiance), dot(yb, radiance), dot(zb, radiance)};
    return (xyzc * 300.0f) / (106.856895f * 4.0f);
}

}  // namespace crt
