// crt_wavefront.hip -- the wavefront form of the path-trace pass for gfx950.
//
// Same computation as ComputeShader.wgsl `main` (:77-117) and bit-identical
// results, but split so that the pointer-chasing part runs alone:
//
//   k_wf_shade   one thread per path slot (streaming, SoA, 16-byte coalesced):
//                consumes last iteration's hit / visibility, runs the shading
//                half of path_trace (:141-292), emits the next extension ray
//                and (diffuse hits) a shadow ray, finishes paths into the
//                staging buffer and re-arms dead slots with the next
//                (sample, pixel) work item -- so the pool stays full.  Active
//                rays are compacted into two lists with __ballot/__popcll
//                prefix sums and ONE atomic per wave per list.
//   k_wf_trace   persistent waves; each lane walks the BVH2 for one ray at a
//                time (stack in LDS, [level][lane]); lanes that finish are
//                refilled from a per-wave chunk of the list (one atomic per
//                chunk), so lanes do not wait for the slowest ray of a wave.
//                Shadow rays run any-hit against the t of the light's own
//                primitive (found in k_wf_shade), which is exactly the
//                reference's "closest hit == the light" test (:697-705).
//   k_wf_resolve per pixel: accum += sample_1 + sample_2 + ... in sample order
//                (:108), then the colour pipeline (:110-115).
//
// RNG call order per path is the reference's: the shadow trace
// consumes no random numbers, so a diffuse bounce can draw its light, hemisphere
// and roulette numbers in one go.
#include "crt_shade.h"
#include <algorithm>

namespace crt {

#ifndef CRT_WF_STACK
#define CRT_WF_STACK 32
#endif
#ifndef CRT_WF_REFILL
#define CRT_WF_REFILL 24
#endif
#ifndef CRT_WF_INNER_RUN
#define CRT_WF_INNER_RUN 6
#endif
#ifndef CRT_WF_BURSTS
#define CRT_WF_BURSTS 4
#endif
#ifndef CRT_WF_BVH4
#define CRT_WF_BVH4 1
#endif
#ifndef CRT_WF_LEAF_AT
#define CRT_WF_LEAF_AT 32
#endif
#ifndef CRT_WF_SPEC
#define CRT_WF_SPEC 1
#endif
#ifndef CRT_WF_PATCH_RECOMPUTE
#define CRT_WF_PATCH_RECOMPUTE 20      /* k_wf_trace2 recomputes a patch's fourth record part where there are more than 2^this primitives per patch */
#endif
#ifndef CRT_WF_SHADE_BLOCK
#define CRT_WF_SHADE_BLOCK 64
#endif
#ifndef CRT_WF_SHADE_MIN_WAVES
#define CRT_WF_SHADE_MIN_WAVES 1
#endif
#ifndef CRT_WF_MIN_WAVES
#define CRT_WF_MIN_WAVES 1
#endif
constexpr int kWfStack = CRT_WF_STACK;      // LDS stack entries per lane (BVH depth is capped by the builder)
constexpr int kNoNode = 0x7FFFFFFF;          // "no node left to walk" (inner ids are smaller, leaf references negative)
constexpr int kTraceChunk = 128;            // list entries a wave reserves per atomic
constexpr int kRefillAt = CRT_WF_REFILL;    // refill when at least this many lanes are idle

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Streams that are touched once per iteration (the pool's per-slot state, ray lists and records, staging) are loaded and
// stored NON-TEMPORALLY: a shade launch moves ~1 GB through an XCD's 4 MB of L2 while the other pipe's traversal kernel
// lives on the scene's nodes and triangles staying there.
#ifndef CRT_WF_NT
#define CRT_WF_NT 1
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#if CRT_WF_NT
__device__ __forceinline__ float4 ldnt(const float4 *p) { const v4f v = __builtin_nontemporal_load((const v4f *)p); return float4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ uint4 ldnt(const uint4 *p) { const v4u v = __builtin_nontemporal_load((const v4u *)p); return uint4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float2 ldnt(const float2 *p) { const v2f v = __builtin_nontemporal_load((const v2f *)p); return float2{v.x, v.y}; }
__device__ __forceinline__ uint32_t ldnt(const uint32_t *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(float4 *p, float4 v) { __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, (v4f *)p); }
__device__ __forceinline__ void stnt(uint4 *p, uint4 v) { __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, (v4u *)p); }
__device__ __forceinline__ void stnt(float2 *p, float2 v) { __builtin_nontemporal_store(v2f{v.x, v.y}, (v2f *)p); }
__device__ __forceinline__ void stnt(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stnt(uint2 *p, uint2 v) { __builtin_nontemporal_store(v2u{v.x, v.y}, (v2u *)p); }
#else
template <class T> __device__ __forceinline__ T ldnt(const T *p) { return *p; }
template <class T> __device__ __forceinline__ void stnt(T *p, T v) { *p = v; }
#endif
__device__ __forceinline__ uint32_t prefix_popc(unsigned long long mask, uint32_t lane)
{
    return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// The reference's own loop (:503-518) for a non-finite ray, out of line and taking only
// values (a by-reference scene would force the kernel-argument struct into scratch).
__device__ __noinline__ float2 resolve_nonfinite(const float4 *prim, const float4 *primD, const uint32_t *slot_of_index,
                                                 uint32_t nprim, float hit_pad, float ox, float oy, float oz, float dx,
                                                 float dy, float dz, uint32_t exclude)
{
    DevScene S{};
    S.prim = prim; S.primD = primD; S.slot_of_index = slot_of_index; S.nprim = nprim; S.hit_pad = hit_pad;
    float tm = CRT_INFINITY;
    uint32_t bi = kNoHit, bs = kNoHit;
    const f3 o = f3{ox, oy, oz}, d = f3{dx, dy, dz};
    for (uint32_t i = 0; i < nprim; i++) hit_test<true>(S, slot_of_index[i], o, d, exclude, 0.001f, tm, bi, bs);
    return float2{tm, bits_f(bs)};
}

// ------------------------------------------------------------------ shade
struct PathRegs {
    f3 ray_o, ray_d;
    f4 beta, radiance;
    Rng rng;
    float last_pdf, etaScale;
    uint32_t exclude, work, flags;
};

__device__ __forceinline__ void wavelengths_of(uint32_t lambda, uint32_t wl[4])
{
    wl[0] = lambda; wl[1] = (lambda + 4u) % kNLambda; wl[2] = (lambda + 8u) % kNLambda;
    wl[3] = (lambda + 12u) % kNLambda;                           // :321
}

// A light as the NEE step needs it: its record (data1/2/3 + emission index, primitive index, 1/area) and the record of
// its own primitive.  UNIFORM (scenes with ONE light -- the reference's cornell box and everything built around it): every
// address is the same for all lanes, so the loads are scalar loads from the constant address space (the scene is read-only
// during a trace), issued at the top of the kernel under the slot streams' round trip -- instead of a chain of four
// dependent per-lane fetches (light -> slot of its primitive -> the primitive's record -> 1/area for the pdf) in the
// middle of the shade step, where a wave squeezed in beside the traversal kernel's waves has nothing to hide them behind.
struct LightRec { float4 L0, L1, L2, A, B, C, D; uint32_t slot; float pdf_area; };

template <bool UNIFORM>
__device__ __forceinline__ LightRec load_light(const DevScene &S, uint32_t li)
{
    LightRec r;
    r.A = float4{0.0f, 0.0f, 0.0f, 0.0f}; r.B = r.A; r.C = r.A; r.D = r.A; r.slot = kNoHit; r.pdf_area = 0.0f;
    if (UNIFORM) {
        typedef const __attribute__((address_space(4))) v4f *cptr4;
        typedef const __attribute__((address_space(4))) uint32_t *cptr1;
        auto ld = [](cptr4 p) { const v4f v = *p; return float4{v.x, v.y, v.z, v.w}; };
        const cptr4 lg = (cptr4)S.lights;
        r.L0 = ld(lg); r.L1 = ld(lg + 1); r.L2 = ld(lg + 2);
        const uint32_t include = f_bits(r.L1.w);
        if (include < S.nprim) {
            r.slot = ((cptr1)S.slot_of_index)[include];
            const cptr4 pg = (cptr4)S.prim + 3 * (size_t)r.slot;
            r.A = ld(pg); r.B = ld(pg + 1); r.C = ld(pg + 2); r.D = ld((cptr4)S.primD + r.slot);
            // the light's 1/area as compute_light_pdf() finds it: lights[min(emission index of the primitive, nlight-1)] = lights[0]
            r.pdf_area = r.L2.w;
        }
    } else {
        r.L0 = S.lights[3 * li + 0]; r.L1 = S.lights[3 * li + 1]; r.L2 = S.lights[3 * li + 2];
        const uint32_t include = f_bits(r.L1.w);
        if (include < S.nprim) {
            r.slot = S.slot_of_index[include];
            r.A = S.prim[3 * (size_t)r.slot + 0]; r.B = S.prim[3 * (size_t)r.slot + 1]; r.C = S.prim[3 * (size_t)r.slot + 2];
            r.D = S.primD[r.slot];
            uint32_t l2 = (f_bits(r.A.w) >> 4) & 0x3FFFu;              // compute_light_pdf(): lights[emission index] (sic, Q7), clamped
            if (l2 >= S.nlight) l2 = S.nlight - 1u;
            r.pdf_area = S.lights[3 * l2 + 2].w;
        }
    }
    return r;
}

// compute_light_radiance's tail (:388-400) for a light primitive hit at t_l along the shadow ray: the NEE term before
// the caller's BRDF * beta (:187).
__device__ __forceinline__ f4 nee_term(const DevScene &S, const LightRec &L, f3 pos, f3 ldir, float t_l, float cos_theta,
                                       const f4 spec)
{
    f3 lp, ln; uint32_t lmeta;
    hit_attributes_rec(L.A, L.B, L.C, L.D, pos, ldir, t_l, lp, ln, lmeta);
    f4 le = spec * cos_theta;
    float pdf_l = compute_light_pdf_area(L.pdf_area, S.inv_nlight, lp, ln, pos, ldir);
    float pdf_b = cos_theta / CRT_PI;
    float weight_l = power_heuristic(1.0f, pdf_l, 1.0f, pdf_b);
    return (le * weight_l) / pdf_l;
}

struct ShadeCnt { uint32_t rays = 0, bounces = 0, shadow = 0, hits = 0, paths = 0, prims = 0, walk = 0; };
struct ShadowOut { bool emit; f3 d; float t_l; uint32_t l_index, l_slot; };

// compute_light_radiance (:379-408) from the sampled point on the light to the visibility test: intersects the light's own
// primitive (shadow_intersect's "closest hit == the light", :697-705, starts there), and either adds what can be decided
// without a walk to `radiance` or stores the NEE term and describes the shadow ray to trace.
template <bool COUNT, bool FINISH>
__device__ __forceinline__ ShadowOut light_sample(const WfParams &P, const LightRec &L, uint32_t slot, f3 pos, f3 nrm, uint32_t b_index,
                                                  float u, float v2, f4 brdf, f4 beta, f4 &radiance, bool &rad_dirty, const uint32_t wl[4], ShadeCnt &cn)
{
    const DevScene &S = P.sc;
    ShadowOut out{false, f3{0.0f, 0.0f, 0.0f}, 0.0f, 0u, 0u};
    f3 pl = (xyz(L.L0) + xyz(L.L1) * u) + xyz(L.L2) * v2;
    f3 ldir = normalize(pl - pos);
    const uint32_t include = f_bits(L.L1.w);
    if (COUNT) { cn.rays++; cn.shadow++; }
    // shadow_intersect (:697-705): the light's own primitive first
    float t_l = CRT_INFINITY;
    uint32_t l_index = kNoHit, l_slot = kNoHit;
    const bool sh_finite = finite3(ldir) && finite3(pos);
    if (include < S.nprim && sh_finite) {
        hit_test_rec<false>(L.A, L.B, L.C, L.D, S.hit_pad, L.slot, pos, ldir, b_index, 0.001f, t_l, l_index, l_slot);
        if (COUNT) cn.prims++;
    }
    // cos_theta == 0 (the light sample is behind the surface): le = spec*0 is exactly 0, so the
    // NEE term (:400) is exactly +0 whatever the visibility -- adding it changes nothing, and
    // the shadow ray need not be walked (weight and pdf are finite: abs_cos >= 1e-5, :366).
    const float cos_theta = max_(0.0f, dot(nrm, ldir));
    if (!sh_finite) {
        // A non-finite shadow ray (the light sample coincides with the hit point, or a light record
        // with non-finite coordinates): normalize() has put a NaN into its direction, and under the
        // reference's reject-form tests (:546,:557,:566,:605,:609) a NaN passes every test of every
        // patch and sphere (never this project's triangles), so "the closest hit" of shadow_intersect
        // is simply the LAST patch / sphere of the array that is not excluded -- the loop's result in
        // closed form (S.nf_last, found at upload).  If that is the light (:700) the NEE term is added:
        // le = spec * max(0, NaN) = 0 and pdf_l = NaN make it NaN in every wavelength (:393-400).
        const uint32_t w = S.nf_last[0] != b_index ? S.nf_last[0] : S.nf_last[1];
        if (COUNT) cn.prims += S.nprim;
        if (w != kNoHit && w == include) {
            if (COUNT) cn.hits++;
            const float qn = bits_f(0x7FC00000u);
            radiance = radiance + f4{qn, qn, qn, qn};
            rad_dirty = true;
        }
    } else if (l_slot != kNoHit && cos_theta > 0.0f) {
        if (COUNT) cn.hits++;
        const f4 nee = nee_term(S, L, pos, ldir, t_l, cos_theta, sample_spectrum(S, f_bits(L.L0.w), wl));
        f4 c = (brdf * nee) * beta;          // added to radiance iff the light is visible
#ifndef CRT_WHATIF_NO_NEE
        stnt(&P.nee[slot], float4{c.x, c.y, c.z, c.w});
#else
        if (c.x == 12345.678f) stnt(&P.nee[slot], float4{c.x, c.y, c.z, c.w});
#endif
        if (FINISH) {                        // (k_wf_finish traces from the slot arrays; the pool's
            P.sh_d[slot] = float4{ldir.x, ldir.y, ldir.z, t_l};   //  traversal kernel from the ray records)
            P.vis[slot] = include;
        }
        out = ShadowOut{true, ldir, t_l, include, l_slot};
        if (COUNT) cn.walk++;
    }
    return out;
}

// What one shade step leaves behind: the path's registers (the slot's next state; the extension ray is R.ray_o / R.ray_d /
// R.exclude) and the shadow ray, if one is emitted (it starts where the extension ray does).
struct ShadeOut {
    bool alive, emit_ext, emit_sh, sh_primary, rad_dirty; uint32_t batch;
    PathRegs R;
    f3 sd; float t_l; uint32_t l_index, l_slot;
};

// Phase clock of the probe build (make EXTRA=-DCRT_WF_PROBE OUT=.../tune_probe.so, tools/shade_probe.py): everything
// issued so far has arrived, then the cycle counter.  Compiles to nothing otherwise.
#ifdef CRT_WF_PROBE
#define CRT_PROBE(tp, k) if (tp) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); (tp)[k] = __builtin_readcyclecounter(); }
#else
#define CRT_PROBE(tp, k)
#endif

// One shade step of one path slot: steps 1-4 of k_wf_shade's description.  FINISH: the slot is
// driven by k_wf_finish (no re-arming from the work queue; the caller traces the emitted rays itself).
template <bool COUNT, bool FINISH>
__device__ __forceinline__ ShadeOut shade_body(const WfParams &P, uint32_t slot, bool in_pool, ShadeCnt &cn, unsigned long long *tp = nullptr)
{
    const DevScene &S = P.sc;
    PathRegs R;
    R.flags = 0;
    LightRec LU{};
    if (S.nlight == 1u) LU = load_light<true>(S, 0u);
    // Every per-slot stream is loaded up front in ONE batch (the kernel is bound by dependent
    // memory round trips, not by bytes: a dead slot's extra 140 B cost nothing next to that).
    uint4 misc = uint4{0, 0, 0, 0}, rs = uint4{0, 0, 0, 0};
    float4 v_ro = float4{0, 0, 0, 0}, v_rd = v_ro, v_beta = v_ro, v_rad = v_ro, v_nee = v_ro;
    float2 h = float2{0.0f, 0.0f};
    uint32_t vis_in = 0;
    if (in_pool) {
        misc = ldnt(&P.misc[slot]); v_ro = ldnt(&P.ray_o[slot]); v_rd = ldnt(&P.ray_d[slot]); v_beta = ldnt(&P.beta[slot]);
        v_rad = ldnt(&P.radiance[slot]); rs = ldnt(&P.rng[slot]); h = ldnt(&P.hit[slot]); vis_in = ldnt(&P.vis[slot]);
#ifndef CRT_WHATIF_NO_NEE                 /* sensitivity probe (wrong image): the pass without the nee stream */
        v_nee = ldnt(&P.nee[slot]);
#endif
    }
    CRT_PROBE(tp, 1)
    R.work = misc.x; R.flags = misc.y; R.last_pdf = bits_f(misc.z); R.etaScale = bits_f(misc.w);
    bool alive = in_pool && (R.flags & kWfAlive);
    bool emit_ext = false, emit_sh = false;
    bool sh_primary = false;                          // ray class: shadow ray of a camera-ray hit
    bool finished = false, rad_dirty = false;
    f3 out_sd = f3{0, 0, 0}; float out_tl = 0.0f; uint32_t out_lindex = 0, out_lslot = 0;   // the shadow ray, if one is emitted
    // A non-finite camera ray of k_wf_gen (the camera itself is not finite) has not been traced: this step only re-emits it,
    // and the write-back below decides it with the reference loop like any other non-finite ray (shade_store) -- the path
    // is shaded one iteration later.  (No second call site of that loop here: it would cost the kernel 10 registers.)
    const bool nan_ray = alive && (R.flags & kWfNanRay) != 0u;
    if (nan_ray) R.flags &= ~kWfNanRay;
    // the hit primitive's whole record, also in one batch (valid only for an alive slot with a hit)
    const uint32_t h_slot = f_bits(h.y);
    const bool has_hit = alive && !nan_ray && !(R.flags & kWfDying) && h_slot != kNoHit;
    float4 hA = float4{0, 0, 0, 0}, hB = hA, hC = hA, hD = hA;
    if (has_hit) { hA = S.prim[3 * (size_t)h_slot + 0]; hB = S.prim[3 * (size_t)h_slot + 1]; hC = S.prim[3 * (size_t)h_slot + 2]; if (S.npatch) hD = S.primD[h_slot]; }
    CRT_PROBE(tp, 2)

    if (alive) {
        R.ray_o = xyz(v_ro); R.exclude = f_bits(v_ro.w);
        R.ray_d = xyz(v_rd);
        R.beta = f4{v_beta.x, v_beta.y, v_beta.z, v_beta.w};
        // (a path's radiance is stored only once something has been added to it: until then the array holds what the
        // slot's previous path left there)
        R.radiance = (R.flags & kWfHasRad) ? f4{v_rad.x, v_rad.y, v_rad.z, v_rad.w} : f4{0.0f, 0.0f, 0.0f, 0.0f};
        R.rng = Rng{rs.x, rs.y, rs.z, rs.w};
        uint32_t wl[4];
        wavelengths_of((R.flags >> kWfLambdaShift) & 0x1FFu, wl);
        uint32_t depth = (R.flags >> kWfDepthShift) & 0xFFu;

        // 1. the NEE term of the previous bounce, now that visibility is known (:187)
        if (R.flags & kWfShadow) {
            if (vis_in == 1u) { R.radiance = R.radiance + f4{v_nee.x, v_nee.y, v_nee.z, v_nee.w}; rad_dirty = true; }
            R.flags &= ~kWfShadow;
        }
        if (R.flags & kWfDying) {
            finished = true;                                     // roulette ended it last iteration (:284-287)
        } else if (nan_ray) {
            emit_ext = true;                                     // (resolved by the write-back; counted as a ray when it was generated)
        } else {
            // 2. the extension ray's closest hit (:135-146)
            const uint32_t b_slot = h_slot;
            if (COUNT) cn.bounces++;
            if (b_slot == kNoHit) {
                finished = true;                                 // :141
            } else {
                f3 pos, nrm;
                const f3 o = R.ray_o, d = R.ray_d;
                // hit attributes from the prefetched record (same operations as hit_attributes())
                const uint32_t meta = f_bits(hA.w);
                pos = ray_at(o, d, h.x);
                if ((meta & 3u) == 1u) {
                    nrm = normalize(pos - xyz(hA));                          // :618
                } else {
                    const f3 n0 = ((meta & 3u) == 0u) ? xyz(hD) : normalize(cross(xyz(hB), xyz(hC)));
                    nrm = (dot(n0, d) > 0.0f) ? -n0 : n0;                    // :541-544
                }
                if (COUNT) cn.hits++;
                const uint32_t b_index = f_bits(hB.w);
                R.exclude = b_index;                             // :146
                const uint32_t material = (meta >> 2) & 3u;
                const uint32_t emission_index = (meta >> 4) & 0x3FFFu;
                const uint32_t reflectance_index = (meta >> 18) & 0x3FFFu;
                if (material == kLight) {                        // :149-164
                    f4 le = sample_spectrum(S, emission_index, wl);
                    if (depth == 0 || (R.flags & kWfSpecular)) {
                        R.radiance = R.radiance + R.beta * le;
                    } else {
                        float pdf_l = compute_light_pdf(S, emission_index, pos, nrm, o, d);
                        float weight_b = power_heuristic(1.0f, R.last_pdf, 1.0f, pdf_l);
                        R.radiance = R.radiance + (le * weight_b) * R.beta;
                    }
                    finished = true;
                } else if (depth >= kMaxDepthPath) {             // :167
                    finished = true;
                } else {
                    if (R.flags & kWfInTrans) {                  // :173-179
                        float distance = length(o - pos);
                        f4 ext = sample_spectrum(S, S.nspectra - 1u, wl);
                        f4 att = f4{exp_(-ext.x * distance), exp_(-ext.y * distance), exp_(-ext.z * distance),
                                    exp_(-ext.w * distance)};
                        R.beta = R.beta * att;
                    }
                    if (material == kDiffuse) {                  // :182-204
                        f4 brdf = sample_spectrum(S, reflectance_index, wl) / CRT_PI;
                        // compute_light_radiance :379-408 up to the visibility test
                        float u0 = rnd(R.rng);
                        uint32_t li = (uint32_t)((float)S.nlight * u0);
                        if (li >= S.nlight) li = S.nlight - 1u;
                        float u = rnd(R.rng);
                        float v2 = rnd(R.rng);
                        // (one light: its records were fetched with scalar loads at the top; else per lane, here)
                        ShadowOut sh;
                        if (S.nlight == 1u) sh = light_sample<COUNT, FINISH>(P, LU, slot, pos, nrm, b_index, u, v2, brdf, R.beta, R.radiance, rad_dirty, wl, cn);
                        else sh = light_sample<COUNT, FINISH>(P, load_light<false>(S, li), slot, pos, nrm, b_index, u, v2, brdf, R.beta, R.radiance, rad_dirty, wl, cn);
                        if (sh.emit) {
                            emit_sh = true;
                            sh_primary = depth == 0u;
                            R.flags |= kWfShadow;
                            out_sd = sh.d; out_tl = sh.t_l; out_lindex = sh.l_index; out_lslot = sh.l_slot;
                        }
                        f3 new_direction = cosine_hemisphere(R.rng, nrm, R.last_pdf);
                        float cos_theta2 = abs_(dot(nrm, new_direction));
                        R.beta = R.beta * ((brdf * cos_theta2) / R.last_pdf);
                        R.ray_o = pos;
                        R.ray_d = new_direction;
                        R.flags &= ~kWfSpecular;
                    } else if (material == kGlass) {             // :208-276
                        const float eta1 = 1.0f, eta2 = 1.5f;
                        float eta = eta1 / eta2;
                        float cos_theta = dot(nrm, d);
                        float reflected = fresnel_s(d, nrm, eta1, eta2);
                        float pr = reflected;
                        float pt = 1.0f - reflected;
                        float u = rnd(R.rng);
                        f3 current_normal = nrm;
                        if (cos_theta > 0.0f) { eta = 1.0f / eta; current_normal = -current_normal; }
                        f3 new_direction;
                        if (u < pr / (pr + pt)) {
                            new_direction = reflect_(d, current_normal);
                        } else {
                            new_direction = normalize(refract_(d, current_normal, eta));
                            R.beta = R.beta * (eta * eta);
                            R.etaScale = R.etaScale / (eta * eta);
                            R.flags ^= kWfInTrans;
                        }
                        R.ray_o = pos;
                        R.flags |= kWfSpecular;
                        R.exclude = 0xFFFFFFFFu;
                        R.ray_d = new_direction;
                    }
                    // Russian roulette :279-289
                    bool rr_break = false;
                    {
                        f4 rbeta = R.beta * R.etaScale;
                        float max_beta_component = max_(rbeta.x, max_(rbeta.y, rbeta.z));
                        if (depth > 1u && max_beta_component < 1.0f) {
                            float qq = max_(0.0f, 1.0f - max_beta_component);
                            if (rnd(R.rng) < qq) rr_break = true;
                            else R.beta = R.beta / (1.0f - qq);
                        }
                    }
                    if (rr_break) {
                        if (emit_sh) R.flags |= kWfDying;        // wait for the NEE visibility, then finish
                        else finished = true;
                    } else {
                        depth++;
                        R.flags = (R.flags & ~(0xFFu << kWfDepthShift)) | (depth << kWfDepthShift);
                        emit_ext = true;
                        if (COUNT) { cn.rays++; cn.walk++; }
                    }
                }
            }
        }
        if (finished) {
            // spectral_to_xyz (:105) into the staging slot of (sample, pixel)
            const uint32_t sample_off = R.work / P.npix_padded, pp = R.work % P.npix_padded;
            const uint32_t tile = pp >> 6, l = pp & 63u;
            const uint32_t lx = (tile % P.tiles_x) * 8u + (l & 7u), ly = (tile / P.tiles_x) * 8u + (l >> 3);
            f3 c = spectral_to_xyz(S, R.radiance, wl);
            stnt(&P.staging[(R.flags >> kWfBatchShift) & (kWfRing - 1u)][(size_t)sample_off * ((size_t)P.tw * P.th) + (size_t)ly * P.tw + lx], float4{c.x, c.y, c.z, 0.0f});
            alive = false;
            if (COUNT) cn.paths++;
        }
    }

    CRT_PROBE(tp, 3)
    if (!alive) R.flags = 0;
    ShadeOut so;
    so.alive = alive; so.emit_ext = emit_ext; so.emit_sh = emit_sh; so.sh_primary = sh_primary; so.rad_dirty = rad_dirty;
    so.batch = (R.flags >> kWfBatchShift) & (kWfRing - 1u);
    so.R = R; so.sd = out_sd; so.t_l = out_tl; so.l_index = out_lindex; so.l_slot = out_lslot;
    return so;
}

// 4. write the slot back.  Returns 1 when the extension ray was resolved here (a non-finite ray, below).
template <bool COUNT, bool FINISH>
__device__ __forceinline__ uint32_t shade_store(const WfParams &P, uint32_t slot, bool in_pool, ShadeOut &so, ShadeCnt &cn)
{
    const DevScene &S = P.sc;
    PathRegs &R = so.R;
    uint32_t resolved = 0u;
    if (in_pool) {
        // ray_o, beta and misc are written for EVERY slot: a dead slot gets what a fresh path starts with (the eye, no exclusion,
        // throughput 1, last_bounce_pdf = etaScale = 1) here, in the launch's coalesced streams, so that k_wf_gen -- whose
        // stores go to scattered slots, 16 bytes at a time: 2.5 x their size in HBM writes -- only writes what depends on the pixel.
        const float4 vo = so.alive ? float4{R.ray_o.x, R.ray_o.y, R.ray_o.z, bits_f(R.exclude)} : float4{S.cam[9], S.cam[10], S.cam[11], bits_f(0xFFFFFFFFu)};
        stnt(&P.ray_o[slot], vo);
        if (so.alive) {
            // A non-finite ray (e.g. refract at the numerical edge of total reflection) is decided
            // by the reference loop in its own order; do that here and flag the ray as resolved
            // so the traversal kernel stays free of the fallback.
            if (so.emit_ext && (!finite3(R.ray_o) || !finite3(R.ray_d))) {
                P.hit[slot] = resolve_nonfinite(S.prim, S.primD, S.slot_of_index, S.nprim, S.hit_pad, R.ray_o.x, R.ray_o.y, R.ray_o.z,
                                                R.ray_d.x, R.ray_d.y, R.ray_d.z, R.exclude);
                if (COUNT) cn.prims += S.nprim;
                resolved = 1u;
            }
            stnt(&P.ray_d[slot], float4{R.ray_d.x, R.ray_d.y, R.ray_d.z, bits_f(resolved)});
            if (so.rad_dirty || (FINISH && !(R.flags & kWfHasRad))) {   // (k_wf_finish re-reads it every step: stored once)
                stnt(&P.radiance[slot], float4{R.radiance.x, R.radiance.y, R.radiance.z, R.radiance.w});
                R.flags |= kWfHasRad;
            }
            stnt(&P.rng[slot], uint4{R.rng.x, R.rng.y, R.rng.z, R.rng.w});
#ifdef CRT_WHATIF_EXTRA_STREAM          /* sensitivity probe: 16 B more read and written per live slot (an array the pool does not use) */
            if (!FINISH) { const float4 x = ldnt(&P.sh_d[slot]); stnt(&P.sh_d[slot], float4{x.y, x.x, x.w, x.z}); }
#endif
        }
        stnt(&P.beta[slot], so.alive ? float4{R.beta.x, R.beta.y, R.beta.z, R.beta.w} : float4{1.0f, 1.0f, 1.0f, 1.0f});
        stnt(&P.misc[slot], so.alive ? uint4{R.work, R.flags, f_bits(R.last_pdf), f_bits(R.etaScale)} : uint4{0u, 0u, f_bits(1.0f), f_bits(1.0f)});
    }
    return resolved;
}

// What the host's driver needs to know about iteration it_end - 1 of this pipe (lane i looks at shard i of
// everything), written by the first wave of the NEXT shade launch straight into a pinned host record: it_end goes last,
// behind a system-scope fence, and the host polls for it -- no kernel, copy or event on the pipe's critical path.
__device__ __noinline__ void write_status(const WfCtl *ctl, const WfWorkQ *wq, uint32_t it_end, WfStatus *out)
{
    // (out of line and in groups of eight batch ids: inlined and fully unrolled it would cost every wave of the shade
    // kernel a third of its occupancy in registers)
    const uint32_t lane = lane_id();
    const WfShard &sh = ctl->shard[(it_end - 1u) & 3u][lane];
    const uint32_t mine = sh.n[0] + sh.n[1] + sh.n[2] + sh.n[3];
    unsigned long long rays = mine;
    uint32_t bound = mine;
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_xor(rays, off, 64);
        bound = max(bound, (uint32_t)__shfl_xor((int)bound, off, 64));
    }
    if (lane == 0) { out->dropped = ctl->dropped; out->bound = bound; out->pad_ = 0; out->rays = rays; }
#pragma unroll 1
    for (uint32_t g = 0; g < kWfRing; g += 8u) {
        unsigned long long cur[8];
        uint32_t alive[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {                      // the group's loads first: one round trip, not eight
            cur[k] = __hip_atomic_load(&wq[g + k].work[lane].cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            alive[k] = sh.alive[g + k];
        }
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {
            const uint32_t b = g + k;
            const uint32_t wps = wq[b].work_per_shard;
            const unsigned long long wtot = wq[b].work_total, lo = (unsigned long long)lane * wps;
            const unsigned long long size = lo < wtot ? min((unsigned long long)wps, wtot - lo) : 0ull;
            unsigned long long cons = min(cur[k], size);
            const unsigned long long left = __ballot(cur[k] < size);
            uint32_t al = alive[k];
            for (int off = 32; off > 0; off >>= 1) {
                al += (uint32_t)__shfl_xor((int)al, off, 64);
                cons += __shfl_xor(cons, off, 64);
            }
            if (lane == 0) { out->alive[b] = al; out->consumed[b] = cons; out->left[b] = left ? 1u : 0u; }
        }
    }
    if (lane == 0) {
        __threadfence_system();
        __hip_atomic_store(&out->it_end, it_end, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(CRT_WF_SHADE_BLOCK, CRT_WF_SHADE_MIN_WAVES) void k_wf_shade(const WfParams P, uint32_t it)
{
    const uint32_t ring = it & 3u, lbuf = it & 1u;
    WfCtl *ctl = P.ctl;
    static_assert(kWfShards == 64, "one lane per shard");
    if (blockIdx.x == 0 && threadIdx.x < 64u && P.status_out != nullptr) write_status(P.ctl, P.wq, it, P.status_out);   // of the previous iteration
    if (blockIdx.x == 0 && threadIdx.x < kWfShards) {            // arm the next iteration's counters
        WfShard &nx = ctl->shard[(it + 1u) & 3u][threadIdx.x];   // (ring it-1 is still read in tail mode)
        nx.n[0] = 0; nx.n[1] = 0; nx.n[2] = 0; nx.n[3] = 0; nx.cur = 0; nx.n_dead = 0;
        for (uint32_t b = 0; b < kWfRing; b++) nx.alive[b] = 0;
    }
    uint32_t slot, my_shard;
    bool in_pool;
    if (P.tail_bound == 0u) {
        const uint32_t local = blockIdx.x * (uint32_t)CRT_WF_SHADE_BLOCK + threadIdx.x;
        slot = P.slot_base + local;
        my_shard = blockIdx.x % kWfShards;
        in_pool = local < P.P;
    } else {
        // Tail mode (no work left, few paths alive): every alive slot listed a ray last iteration,
        // so walk those lists instead of the whole pool.  Per shard: threads [c*bound,(c+1)*bound)
        // take list class c; a shadow-list entry whose slot also listed an extension ray carries
        // kWfListAlsoExt and is skipped here.  (The mark is made when the entry is written: the slot's
        // flags cannot be consulted, the thread that holds the slot's extension entry rewrites them in
        // this very launch.)
        const uint32_t bps = (4u * P.tail_bound + (uint32_t)CRT_WF_SHADE_BLOCK - 1u) / (uint32_t)CRT_WF_SHADE_BLOCK;
        my_shard = blockIdx.x / bps;
        const uint32_t j = (blockIdx.x % bps) * (uint32_t)CRT_WF_SHADE_BLOCK + threadIdx.x;
        const WfShard &pv = ctl->shard[(it + 3u) & 3u][my_shard];
        const size_t region = (size_t)my_shard * P.list_cap;
        slot = 0; in_pool = false;
        const uint32_t cls = j / P.tail_bound, e = j % P.tail_bound;
        if (cls < 4u && e < pv.n[cls]) {
            // (the records are the lists: an extension ray's slot is in recB.w -- its mark there means "resolved", the
            // slot is still this thread's -- a shadow ray's in recC.x)
            const size_t g = (size_t)((lbuf ^ 1u) * 4u + cls) * ((size_t)P.list_cap * kWfShards) + region + e;
            const uint32_t entry = cls < 2u ? f_bits(ldnt(&P.recB[g]).w) : ldnt(&P.recC[g]).x;
            slot = entry & kWfListSlot;
            in_pool = cls < 2u || !(entry & kWfListAlsoExt);     // else reached through its extension ray
        }
    }
    if (P.evict_mask) {
        // Move the (few) paths of the batches named by evict_mask out of the pool: their rays of the last
        // iteration are traced, so the slot state is complete; k_wf_finish continues them from the side pool.
        // The slot is then dead and is listed for k_wf_gen below like any other.
        uint4 misc = uint4{0, 0, 0, 0};
        if (in_pool) misc = P.misc[slot];
        const uint32_t par = (misc.y >> kWfBatchShift) & (kWfRing - 1u);
        const bool go = in_pool && (misc.y & kWfAlive) && ((P.evict_mask >> par) & 1u);
        for (uint32_t b = 0; b < kWfRing; b++) {
            if (!((P.evict_mask >> b) & 1u)) continue;
            const unsigned long long m = __ballot(go && par == b);
            if (!m) continue;
            uint32_t base = 0;
            if (lane_id() == 0) base = atomicAdd(&ctl->side_count[b], (uint32_t)__popcll(m));
            base = __shfl(base, 0, 64);
            const uint32_t idx = base + prefix_popc(m, lane_id());
            if (go && par == b && idx >= kWfSideCap) atomicAdd(&ctl->dropped, 1u);   // must not happen: the host checks
            if (go && par == b && idx < kWfSideCap) {            // (the host asks only when everything fits)
                const uint32_t d = P.side_base[b] + idx;
                P.ray_o[d] = P.ray_o[slot]; P.ray_d[d] = P.ray_d[slot]; P.sh_d[d] = P.sh_d[slot]; P.beta[d] = P.beta[slot];
                P.radiance[d] = P.radiance[slot]; P.nee[d] = P.nee[slot]; P.rng[d] = P.rng[slot]; P.hit[d] = P.hit[slot];
                P.vis[d] = P.vis[slot]; P.misc[d] = misc;
                P.misc[slot] = uint4{0, 0, 0, 0};
            }
        }
    }
    ShadeCnt cn;
#ifdef CRT_WF_PROBE
    // one wave in 256 clocks its phases into counters 8..14 (more would serialise on the counters' lines)
    unsigned long long tpa[8];
    unsigned long long *tp = (!COUNT && P.tail_bound == 0u && (blockIdx.x & 255u) == 7u) ? tpa : nullptr;
    tpa[0] = __builtin_readcyclecounter();
#else
    unsigned long long *tp = nullptr;
#endif
    ShadeOut so = shade_body<COUNT, false>(P, slot, in_pool, cn, tp);
    CRT_PROBE(tp, 4)
    const bool emit_ext = so.emit_ext, emit_sh = so.emit_sh, sh_primary = so.sh_primary;
    {
        const uint32_t lane = lane_id();
        // Ray classes keep like with like in the traversal kernel: camera rays (class 0, listed by k_wf_gen),
        // bounce rays, shadow rays of camera-ray hits (coherent origins), other shadow rays.  Dead slots go to
        // the shard's dead list, from which k_wf_gen re-arms them by whole waves.
        const bool cl2 = emit_sh && sh_primary, cl3 = emit_sh && !sh_primary;
        const bool dead = in_pool && !so.alive && P.rearm != 0u;
        const unsigned long long m1 = __ballot(emit_ext), m2 = __ballot(cl2), m3 = __ballot(cl3), md = __ballot(dead);
        WfShard &sh = ctl->shard[ring][my_shard];
        // The list positions first: the atomics' round trip runs under the write-back of the slot below.
        uint32_t b1 = 0, b2 = 0, b3 = 0, bd = 0;
        if (lane == 0) {
            if (m1) b1 = atomicAdd(&sh.n[1], (uint32_t)__popcll(m1));
            if (m2) b2 = atomicAdd(&sh.n[2], (uint32_t)__popcll(m2));
            if (m3) b3 = atomicAdd(&sh.n[3], (uint32_t)__popcll(m3));
            if (md) bd = atomicAdd(&sh.n_dead, (uint32_t)__popcll(md));
        }
        if (P.count_alive) {
            // paths still in the pool, per batch id: the host retires a batch (eviction of its last paths, resolve)
            // by these counts (k_wf_gen adds the paths it starts).  A wave holds paths of one to three batches as a
            // rule: one ballot + atomic for each.
            unsigned long long am = __ballot(so.alive);
            while (am) {
                const uint32_t b = (uint32_t)__shfl((int)so.batch, __ffsll((long long)am) - 1, 64);
                const unsigned long long m = __ballot(so.alive && so.batch == b);
                if (lane == 0) atomicAdd(&sh.alive[b], (uint32_t)__popcll(m));
                am &= ~m;
            }
        }
        const uint32_t resolved = shade_store<COUNT, false>(P, slot, in_pool, so, cn);
        CRT_PROBE(tp, 5)
        b1 = __shfl(b1, 0, 64); b2 = __shfl(b2, 0, 64); b3 = __shfl(b3, 0, 64); bd = __shfl(bd, 0, 64);
        const size_t region = (size_t)my_shard * P.list_cap;
        // The rays as compacted records in list order: the traversal kernel streams them -- one coalesced round trip
        // per refill instead of list entry -> slot -> ray_o / ray_d (-> light index -> its slot), three to four
        // dependent ones through HBM-resident pool arrays.
        const size_t cls_stride = (size_t)P.list_cap * kWfShards;
        const PathRegs &R = so.R;
        if (emit_ext) {
            const size_t g = (size_t)(lbuf * 4u + 1u) * cls_stride + region + b1 + prefix_popc(m1, lane);
            stnt(&P.recA[g], float4{R.ray_o.x, R.ray_o.y, R.ray_o.z, bits_f(R.exclude)});
            stnt(&P.recB[g], float4{R.ray_d.x, R.ray_d.y, R.ray_d.z, bits_f(slot | (resolved ? kWfListAlsoExt : 0u))});
        }
        if (emit_sh) {
            const uint32_t c = cl2 ? 2u : 3u;
            const size_t g = (size_t)(lbuf * 4u + c) * cls_stride + region + (cl2 ? b2 + prefix_popc(m2, lane) : b3 + prefix_popc(m3, lane));
            stnt(&P.recA[g], float4{R.ray_o.x, R.ray_o.y, R.ray_o.z, bits_f(R.exclude)});
            stnt(&P.recB[g], float4{so.sd.x, so.sd.y, so.sd.z, so.t_l});
            stnt(&P.recC[g], uint4{slot | (emit_ext ? kWfListAlsoExt : 0u), so.l_index, so.l_slot, 0u});
        }
        if (dead) stnt(&P.dead[region + bd + prefix_popc(md, lane)], slot);
    }
    if (COUNT) {
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_RAYS, cn.rays);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_BOUNCES, cn.bounces);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_SHADOW, cn.shadow);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_HITS, cn.hits);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_PATHS, cn.paths);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_PRIMS, cn.prims);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_WALKED, cn.walk);
    }
#ifdef CRT_WF_PROBE
    CRT_PROBE(tp, 6)
    if (tp && lane_id() == 0) {
        for (int k = 0; k < 6; k++) atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 8 + k, tp[k + 1] - tp[k]);
        atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 14, 1ull);
    }
#endif
}

// ------------------------------------------------------------------ generate
// Re-arm dead slots with the next (sample, pixel) work items -- by WHOLE WAVES: k_wf_shade lists the slots that are dead
// after its launch (per shard, ballot-compacted), and here wave (shard, j) takes chunks j, j + gen_blocks, ... of 64 such
// slots and, for each, 64 CONSECUTIVE work items = the 64 pixels of one 8x8 tile of one sample: one queue atomic per
// wave, the sample / tile decode once per wave, a coalesced seed-table read, and every lane on the same code (inside the
// shade kernel the same work ran with a fifth of the lanes on, after up to 6 x 32 attempts at the queues, with two
// integer divisions per lane: a quarter of a shade wave's time).  A queue's work range is split into kWfShards sub-ranges
// with their own cursors, all advanced in multiples of 64, and several queues (batches) can be listed, oldest first: a
// wave takes the work for all of its chunks with ONE atomic from its OWN shard of the oldest listed queue that holds
// some (the shards of a queue drain at the same rate, so at a batch boundary a wave simply goes on with the next
// batch's queue), and looks at the other shards' cursors with one wave-wide load only when all of its own are dry.
// Fewer than 64 dead slots at the end of a shard's list stay dead until the next launch lists them again.  The camera
// rays are ray class 0 of the iteration's lists.
template <bool COUNT>
__global__ __launch_bounds__(64) void k_wf_gen(const WfParams P, uint32_t it)
{
    const DevScene &S = P.sc;
    const uint32_t ring = it & 3u, lbuf = it & 1u;
    WfCtl *ctl = P.ctl;
    const uint32_t lane = lane_id();
    const uint32_t my_shard = blockIdx.x % kWfShards, j = blockIdx.x / kWfShards;
    WfShard &sh = ctl->shard[ring][my_shard];
    const uint32_t n_chunks = min(sh.n_dead, P.list_cap) / 64u;          // (final: the shade launch is through)
    if (j >= n_chunks) return;
    const uint32_t mine = (n_chunks - j + P.gen_blocks - 1u) / P.gen_blocks;     // this wave's chunks: j, j + gen_blocks, ...
    const size_t region = (size_t)my_shard * P.list_cap;
    const size_t cls_stride = (size_t)P.list_cap * kWfShards;
    // ---- the work for ALL of this wave's chunks in as few round trips as possible (the kernel sits between the shade
    //      and the traversal launch of its pipe and is bound by dependent round trips, not by arithmetic): up to two
    //      segments of consecutive work items -- the wave's own shard of the listed queues, oldest first, or one other
    //      shard when all of its own are dry.  (Ranges, totals and steps are multiples of 64.)  What finds no work stays dead.
    uint32_t seg_q[2] = {0, 0}, seg_n[2] = {0, 0}, nseg = 0, want = mine;
    uint32_t seg_w[2] = {0, 0};                                          // (work ids fit 32 bits: wf_batch_cap)
    for (uint32_t si = 0; si < P.seg_n && want > 0u && nseg < 2u; si++) {
        const uint32_t sg = P.seg_order[si];
        WfWorkQ *wq = P.wq + sg;
        const uint32_t wps = P.seg[sg].work_per_shard;
        const unsigned long long wtot = P.seg[sg].work_total, lo = (unsigned long long)my_shard * wps;
        const uint32_t size = lo < wtot ? (uint32_t)min((unsigned long long)wps, wtot - lo) : 0u;
        // (one uniform load of the cursor: no atomic on a shard that is already dry)
        if (__hip_atomic_load(&wq->work[my_shard].cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= size) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&wq->work[my_shard].cur, want * 64u);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= size) continue;
        const uint32_t got = min(want, (size - base) / 64u);
        seg_q[nseg] = sg; seg_w[nseg] = (uint32_t)lo + base; seg_n[nseg] = got; nseg++;
        want -= got;
    }
    for (uint32_t si = 0; si < P.seg_n && nseg == 0u; si++) {            // every shard of every listed queue, one try each
        const uint32_t sg = P.seg_order[si];
        WfWorkQ *wq = P.wq + sg;
        if (__hip_atomic_load(&wq->work_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) continue;
        const uint32_t wps = P.seg[sg].work_per_shard;
        const unsigned long long wtot = P.seg[sg].work_total, lo_l = (unsigned long long)lane * wps;
        const uint32_t size_l = lo_l < wtot ? (uint32_t)min((unsigned long long)wps, wtot - lo_l) : 0u;
        const uint32_t cur_l = __hip_atomic_load(&wq->work[lane].cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long avail = __ballot(cur_l < size_l);
        if (!avail) {                                                    // every shard is exhausted
            if (lane == 0) __hip_atomic_store(&wq->work_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        const uint32_t rot = my_shard & 63u;
        const unsigned long long rmask = rot ? ((avail >> rot) | (avail << (64u - rot))) : avail;
        const uint32_t s_pick = ((uint32_t)(__ffsll((long long)rmask) - 1) + rot) & 63u;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&wq->work[s_pick].cur, want * 64u);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t size_s = (uint32_t)__shfl((int)size_l, (int)s_pick, 64);
        if (base >= size_s) continue;
        const uint32_t got = min(want, (size_s - base) / 64u);
        seg_q[0] = sg; seg_w[0] = s_pick * wps + base; seg_n[0] = got; nseg = 1;
        want -= got;
    }
    const uint32_t total = seg_n[0] + seg_n[1];
    if (total == 0u) return;                                             // no work anywhere: the slots stay dead
    // ---- list positions of the whole wave's camera rays: one atomic.  Every item of an 8x8 tile inside the frame
    //      yields a path; an item outside a ragged tile is consumed without one (its slot stays dead).
    const bool ragged = ((P.tw | P.th) & 7u) != 0u;
    uint32_t n_valid[2] = {seg_n[0] * 64u, seg_n[1] * 64u};
    if (ragged) {
        for (uint32_t g = 0; g < 2u; g++) {
            n_valid[g] = 0;
            for (uint32_t k = 0; k < seg_n[g]; k++) {
                const uint32_t tile = ((seg_w[g] + 64u * k) % P.npix_padded) >> 6;
                const uint32_t lx = (tile % P.tiles_x) * 8u + (lane & 7u), ly = (tile / P.tiles_x) * 8u + (lane >> 3);
                n_valid[g] += (uint32_t)__popcll(__ballot(lx < P.tw && ly < P.th));
            }
        }
    }
    uint32_t b0 = 0;
    if (lane == 0) {
        b0 = atomicAdd(&sh.n[0], n_valid[0] + n_valid[1]);
        if (P.count_alive) {
            if (n_valid[0]) atomicAdd(&sh.alive[seg_q[0]], n_valid[0]);
            if (n_valid[1]) atomicAdd(&sh.alive[seg_q[1]], n_valid[1]);
        }
    }
    uint32_t c_rays = 0, pos = 0;                                        // pos: camera rays of this wave listed so far
    bool have_b0 = false;
    for (uint32_t k = 0; k < total; k++) {
        const uint32_t g = k < seg_n[0] ? 0u : 1u, kk = g ? k - seg_n[0] : k;
        const uint32_t sg = seg_q[g];
        const uint32_t w0 = seg_w[g] + 64u * kk;
        const uint32_t sample_off = w0 / P.npix_padded, pp0 = w0 % P.npix_padded;
        const uint32_t tile = pp0 >> 6;
        const uint32_t lx = (tile % P.tiles_x) * 8u + (lane & 7u), ly = (tile / P.tiles_x) * 8u + (lane >> 3);
        const bool valid = lx < P.tw && ly < P.th;
        const unsigned long long mv = __ballot(valid);
        const uint32_t slot = ldnt(&P.dead[region + (size_t)(j + k * P.gen_blocks) * 64u + lane]);
        f3 eye = f3{0.0f, 0.0f, 0.0f}, d = eye;
        uint32_t resolved = 0u;
        if (valid) {
            const uint32_t px = P.x0 + lx, sample = P.seg[sg].first_sample + sample_off;
            const uint32_t py = P.y0 + (ly / P.band) * P.band * P.stride + P.phase * P.band + ly % P.band;
            Rng rng = Rng{py, px * 100u, sample, P.tea[(size_t)ly * P.tw + lx]};             // :98 (tea(px, py*100) from k_wf_tea)
            const float jx = rnd(rng);
            const float fs = ((float)px + ((float)(sample % kGrid) + jx) / (float)kGrid) / (float)S.W;
            const float jy = rnd(rng);
            const float ft = ((float)S.H - (float)py + ((float)(sample % kGrid) + jy) / (float)kGrid) / (float)S.H;
            const f3 llc = f3{S.cam[0], S.cam[1], S.cam[2]}, hor = f3{S.cam[3], S.cam[4], S.cam[5]};
            const f3 ver = f3{S.cam[6], S.cam[7], S.cam[8]};
            eye = f3{S.cam[9], S.cam[10], S.cam[11]};
            d = normalize(((llc + hor * fs) + ver * ft) - eye);
            const float ul = rnd(rng);
            const uint32_t lambda = (uint32_t)(301.0f * ul);                                  // :317-319
            const bool nan_ray = !finite3(eye) || !finite3(d);
            const uint32_t flags = kWfAlive | (lambda << kWfLambdaShift) | (sg << kWfBatchShift) | (nan_ray ? kWfNanRay : 0u);
            // (ray_o = the eye, beta = 1, last_bounce_pdf = etaScale = 1 are in the slot already: k_wf_shade writes them into
            // every dead slot with its coalesced streams)
            stnt(&P.ray_d[slot], float4{d.x, d.y, d.z, bits_f(0u)});
            stnt(&P.rng[slot], uint4{rng.x, rng.y, rng.z, rng.w});
            stnt((uint2 *)&P.misc[slot], uint2{w0 + lane, flags});                   // work, flags
            // (a camera ray is finite unless the camera itself is not.  Such a ray is decided by the reference loop in its own
            // order like any other non-finite ray -- by the NEXT shade step (kWfNanRay): a call to that loop in this kernel would
            // cost it half its occupancy in registers, and the kernel sits between the shade and the traversal launch of its pipe)
            if (!finite3(eye) || !finite3(d)) resolved = 1u;
        }
        if (!have_b0) { b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0); have_b0 = true; }   // (its round trip ran under the first set-up)
        if (valid) {
            const size_t e = (size_t)(lbuf * 4u) * cls_stride + region + b0 + pos + prefix_popc(mv, lane);
            stnt(&P.recA[e], float4{eye.x, eye.y, eye.z, bits_f(0xFFFFFFFFu)});
            stnt(&P.recB[e], float4{d.x, d.y, d.z, bits_f(slot | (resolved ? kWfListAlsoExt : 0u))});
            if (COUNT) c_rays++;
        }
        pos += (uint32_t)__popcll(mv);
    }
    if (COUNT) {
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_RAYS, c_rays);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_WALKED, c_rays);
    }
}

// ------------------------------------------------------------------ trace
// Branch-free Moeller-Trumbore (same operations and order as hit_test's category 2): every
// lane runs the whole test, the accept decision is one predicate, and only an accepted
// candidate takes the (rare) bounding-box acceptance branch.  Early-outs would not save
// anything under SIMT -- the wave runs until its last lane is through.
__device__ __forceinline__ void tri_test(const float4 A, const float4 B, const float4 C, uint32_t slot, f3 o, f3 d,
                                         uint32_t exclude, float t_min, float hit_pad, float &t_max,
                                         uint32_t &b_index, uint32_t &b_slot)
{
    const uint32_t index = f_bits(B.w);
    const f3 v0 = xyz(A), e1 = xyz(B), e2 = xyz(C);
    const f3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    const float inv = 1.0f / det;
    const f3 tvec = o - v0;
    const float u = dot(tvec, pvec) * inv;
    const f3 qvec = cross(tvec, e1);
    const float v = dot(d, qvec) * inv;
    const float t = dot(e2, qvec) * inv;
    const bool ok = (exclude != index) & (det != 0.0f) & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & ((u + v) <= 1.0f) &
                    beats<false>(t, t_min, t_max, index, b_index, b_slot);
    if (ok) {
        const f3 p = ray_at(o, d, t);
        const f3 v1 = v0 + e1, v2 = v0 + e2;
        const bool in = p.x >= min_(v0.x, min_(v1.x, v2.x)) - hit_pad && p.x <= max_(v0.x, max_(v1.x, v2.x)) + hit_pad &&
                        p.y >= min_(v0.y, min_(v1.y, v2.y)) - hit_pad && p.y <= max_(v0.y, max_(v1.y, v2.y)) + hit_pad &&
                        p.z >= min_(v0.z, min_(v1.z, v2.z)) - hit_pad && p.z <= max_(v0.z, max_(v1.z, v2.z)) + hit_pad;
        if (in) { t_max = t; b_index = index; b_slot = slot; }
    }
}

// Pop entry sp of a lane's stack: the LDS part with a DS read, unconditionally (index clamped), and the global overflow
// area under a branch of its own.  (Written as one conditional expression the two became a FLAT load of a selected
// address: every pop went through the texture-address path -- the unit this kernel saturates -- at global-memory
// latency, with a wait for all outstanding loads behind it.)
__device__ __forceinline__ int stack_pop(const int *stk, const int *ovf, size_t ovl, int sp)
{
    int v = stk[(sp < kWfStack ? sp : kWfStack - 1) * 64];
    if (sp >= kWfStack) v = *(const volatile int *)(ovf + (size_t)(sp - kWfStack) * ovl);   // (volatile: keeps it apart from the DS read)
    return v;
}

// Persistent waves.  A wave owns a chunk of one shard's ray list at a time; entry i of a shard:
// entries of the four class lists in order (camera rays, bounce rays, shadow rays of camera hits, shadow rays).
// Traversal is "while-while": a bounded run of inner-node steps (lanes that reach a leaf wait,
// cheaply), then one leaf step for every lane that has one -- so the expensive primitive tests
// run with most lanes on.
// QUANT: 0 = plain 4-wide nodes (128 B), 1 = quantised 4-wide (64 B), 2 = quantised 8-wide (128 B)
template <bool COUNT, int QUANT>
__global__ __launch_bounds__(64, CRT_WF_MIN_WAVES) void k_wf_trace(const WfParams P, uint32_t it)
{
    __shared__ int lds_stack[kWfStack * 64];
    WfCtl *ctl = P.ctl;
    // hot arrays as plain locals (keeps them in the global address space: global_load, not flat_load)
    const float4 *__restrict__ nodes = CRT_WF_BVH4 ? P.sc.nodes4 : P.sc.nodes;
    const uint4 *__restrict__ nodesq = QUANT == 2 ? P.sc.nodes8q : P.sc.nodes4q;
    const f3 qscale = f3{P.sc.qscale[0], P.sc.qscale[1], P.sc.qscale[2]}, qbase = f3{P.sc.qbase[0], P.sc.qbase[1], P.sc.qbase[2]};
    const float4 *__restrict__ prim = P.sc.prim;
    const float4 *__restrict__ primD = P.sc.primD;
    uint32_t *__restrict__ g_vis = P.vis;
    float2 *__restrict__ g_hit = P.hit;
    // the rays of this iteration, as compacted records in list order (written by k_wf_shade next to the list entries)
    const size_t cls_stride = (size_t)P.list_cap * kWfShards;
    const float4 *__restrict__ recA = P.recA + (size_t)((it & 1u) * 4u) * cls_stride;
    const float4 *__restrict__ recB = P.recB + (size_t)((it & 1u) * 4u) * cls_stride;
    const uint4 *__restrict__ recC = P.recC + (size_t)((it & 1u) * 4u) * cls_stride;
    const float hit_pad = P.sc.hit_pad;
    const int root = QUANT == 2 ? P.sc.root8 : CRT_WF_BVH4 ? P.sc.root4 : P.sc.root;
    int *__restrict__ ovf = P.stack_overflow + ((size_t)blockIdx.x * 64 + lane_id());
    const size_t ovl = P.overflow_lanes;
    const uint32_t nprim = P.sc.nprim;
    DevScene S = P.sc;                                         // for the rare patch / sphere tests
    S.prim = prim; S.primD = primD;

    const uint32_t ring = it & 3u;
    const uint32_t lane = lane_id();
    int *stk = lds_stack + lane;
    const float t_min = 0.001f;

    // wave-uniform fetch state: current shard, its ext count, the reserved chunk [pos,end)
    uint32_t cur_shard = blockIdx.x % kWfShards, sh_e0 = 0, sh_e1 = 0, sh_e2 = 0, sh_total = 0;   // class end offsets
    uint32_t chunk_pos = 0, chunk_end = 0;
    bool have_shard = false, exhausted = false;
    {   // start on this block's own shard (its counts are final: written by k_wf_shade); scan only when it is dry
        const WfShard &so = ctl->shard[ring][cur_shard];
        sh_e0 = so.n[0]; sh_e1 = sh_e0 + so.n[1]; sh_e2 = sh_e1 + so.n[2]; sh_total = sh_e2 + so.n[3];
        have_shard = sh_total > 0u;
    }
    bool active = false;
    int pend = 0;                                   // a leaf put aside (CRT_WF_SPEC), 0 = none
    // per-lane ray + traversal state
    f3 o = f3{0, 0, 0}, d = f3{0, 0, 0}, id = f3{0, 0, 0}, oid = f3{0, 0, 0};
    uint32_t excl = 0, slot = 0, b_index = kNoHit, b_slot = kNoHit, b_slot_in = kNoHit;
    float t_max = 0.0f;
    bool shadow = false;
    int node = 0, sp = 0;
    int nx = 0, ny = 0, nz = 0;                      // 0: lo plane is the near one on that axis, 3: hi plane
    uint32_t c_nodes = 0, c_prims = 0;
    uint32_t d_inner_it = 0, d_inner_act = 0, d_leaf_it = 0, d_leaf_act = 0, d_prim_it = 0, d_refill = 0, d_refill_lanes = 0, d_scans = 0;   // lane 0 only

    for (;;) {
        // ---- refill idle lanes from the wave's chunk
        const unsigned long long idle = __ballot(!active);
        const int nidle = __popcll(idle);
        if (!exhausted && nidle >= kRefillAt) {
            // reserve a chunk if the current one is used up (bounded: every pass either gets a
            // chunk, moves to a shard that had work a moment ago, or finds all shards empty)
            for (int guard = 0; chunk_pos == chunk_end && !exhausted && guard < 2 * (int)kWfShards; guard++) {
                if (have_shard) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&ctl->shard[ring][cur_shard].cur, (uint32_t)kTraceChunk);
                    base = __shfl(base, 0, 64);
                    if (base < sh_total) { chunk_pos = base; chunk_end = min(base + (uint32_t)kTraceChunk, sh_total); break; }
                    have_shard = false;
                }
                // look at every shard at once: lane i loads shard i
                if (COUNT) d_scans++;
                const WfShard &sl = ctl->shard[ring][lane % kWfShards];
                const uint32_t e0 = sl.n[0], e1 = e0 + sl.n[1], e2 = e1 + sl.n[2], tot = e2 + sl.n[3];
                const uint32_t cur_l = __hip_atomic_load(&sl.cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long avail = __ballot(cur_l < tot);
                if (!avail) { exhausted = true; break; }
                const uint32_t rot = cur_shard & 63u;
                const unsigned long long rmask = rot ? ((avail >> rot) | (avail << (64u - rot))) : avail;
                cur_shard = ((uint32_t)(__ffsll((long long)rmask) - 1) + rot) & 63u;
                sh_e0 = __shfl(e0, (int)cur_shard, 64); sh_e1 = __shfl(e1, (int)cur_shard, 64);
                sh_e2 = __shfl(e2, (int)cur_shard, 64); sh_total = __shfl(tot, (int)cur_shard, 64);
                have_shard = true;
            }
            if (chunk_pos < chunk_end) {
                const uint32_t give = min((uint32_t)nidle, chunk_end - chunk_pos);
                if (COUNT) { d_refill++; d_refill_lanes += give; }
                const uint32_t my = prefix_popc(idle, lane);
                if (!active && my < give) {
                    const uint32_t idx = chunk_pos + my;
                    const size_t region = (size_t)cur_shard * P.list_cap;
                    shadow = idx >= sh_e1;
                    const size_t g = idx < sh_e0 ? region + idx : idx < sh_e1 ? cls_stride + region + (idx - sh_e0)
                                   : idx < sh_e2 ? 2u * cls_stride + region + (idx - sh_e1) : 3u * cls_stride + region + (idx - sh_e2);
                    // set up the ray: the whole record in one round trip
                    const float4 ro = ldnt(&recA[g]), rd = ldnt(&recB[g]);
                    uint4 rc = uint4{0, 0, 0, 0};
                    if (shadow) rc = ldnt(&recC[g]);
                    o = xyz(ro); excl = f_bits(ro.w);
                    d = xyz(rd);
                    active = true;
                    if (shadow) {
                        slot = rc.x & kWfListSlot; t_max = rd.w;    // (the mark: the slot also listed an extension ray -- tail mode's business)
                        b_index = rc.y;                           // the light's primitive index and slot
                        b_slot = rc.z;
                    } else {
                        slot = f_bits(rd.w) & kWfListSlot;
                        t_max = CRT_INFINITY; b_index = kNoHit; b_slot = kNoHit;
                        if (f_bits(rd.w) & kWfListAlsoExt) active = false;   // non-finite ray, already resolved by k_wf_shade
                        else if (nprim == 0u) { g_hit[slot] = float2{t_max, bits_f(kNoHit)}; active = false; }
                    }
                    b_slot_in = b_slot;
                    node = root; sp = 0; pend = 0;
                    const float tiny = 1.0e-20f;
                    id.x = 1.0f / (abs_(d.x) > tiny ? d.x : __builtin_copysignf(tiny, d.x));
                    id.y = 1.0f / (abs_(d.y) > tiny ? d.y : __builtin_copysignf(tiny, d.y));
                    id.z = 1.0f / (abs_(d.z) > tiny ? d.z : __builtin_copysignf(tiny, d.z));
                    oid = f3{o.x * id.x, o.y * id.y, o.z * id.z};
                    nx = id.x >= 0.0f ? 0 : 3; ny = id.y >= 0.0f ? 0 : 3; nz = id.z >= 0.0f ? 0 : 3;
                    if (QUANT) {
                        // plane = qbase + q*qscale  =>  t = q*(qscale*id) + (qbase*id - o*id): one fma per plane
                        oid = f3{fma_(qbase.x, id.x, -oid.x), fma_(qbase.y, id.y, -oid.y), fma_(qbase.z, id.z, -oid.z)};
                        id = f3{qscale.x * id.x, qscale.y * id.y, qscale.z * id.z};
                    }
                }
                chunk_pos += give;
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ---- traversal: each pass is either ONE inner-node step for every lane that sits on an inner
        //      node, or ONE leaf for every lane that holds one.  Leaves are postponed until enough
        //      lanes hold one (or nothing else can run), so both code blocks run with many lanes on.
#pragma unroll 1
        for (int pass = 0; pass < 64; pass++) {
#if CRT_WF_SPEC
            // A lane that reaches a leaf puts it aside (one per lane) and goes on with the next node of its stack
            // instead of idling until enough lanes hold a leaf: the closest hit does not depend on the order, only
            // t_max shrinks a little later.
            if (active && node < 0 && pend == 0 && sp > 0) {
                pend = node;
                sp--; node = stack_pop(stk, ovf, ovl, sp);
            }
#endif
            const bool inner = active && node >= 0 && node != kNoNode;
            const bool leaf = active && (node < 0 || pend != 0);
            const int ni = __popcll(__ballot(inner)), nl = __popcll(__ballot(leaf)), nb = __popcll(__ballot(inner || leaf));
            if (nb == 0) break;
            if (!exhausted && 64 - nb >= kRefillAt && pass > 0) break;            // enough idle lanes: refill first
            if (nl >= CRT_WF_LEAF_AT || ni == 0) {
                if (COUNT) {
                    d_leaf_it++; d_leaf_act += (uint32_t)nl;
                    uint32_t mc = leaf ? ((~(uint32_t)(pend != 0 ? pend : node)) & 7u) + 1u : 0u;
                    for (int off = 32; off > 0; off >>= 1) mc = max(mc, (uint32_t)__shfl_xor((int)mc, off, 64));
                    d_prim_it += mc;
                }
                if (leaf) {
                    const bool from_pend = pend != 0;                    // the postponed leaf first
                    const uint32_t enc = ~(uint32_t)(from_pend ? pend : node);
                    const uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
                    for (uint32_t i = 0; i < cnt; i++) {
                        const uint32_t ps = first + i;
                        const float4 A = prim[3 * (size_t)ps + 0], B = prim[3 * (size_t)ps + 1], C = prim[3 * (size_t)ps + 2];
                        if ((f_bits(A.w) & 3u) == 2u) tri_test(A, B, C, ps, o, d, excl, t_min, hit_pad, t_max, b_index, b_slot);
                        else hit_test<false>(S, ps, o, d, excl, t_min, t_max, b_index, b_slot);
                    }
                    if (COUNT) c_prims += cnt;
                    if (from_pend) pend = 0;
                    else if (sp > 0) { sp--; node = stack_pop(stk, ovf, ovl, sp); }
                    else node = kNoNode;
                    bool done = false;
                    if (shadow && b_slot != b_slot_in) done = true;    // any-hit: something beats the light
                    else if (node == kNoNode && pend == 0) done = true;
                    if (done) {
                        if (shadow) stnt(&g_vis[slot], (b_slot == b_slot_in) ? 1u : 0u);
                        else stnt(&g_hit[slot], float2{t_max, bits_f(b_slot)});
                        active = false;
                    }
                }
            } else {
                if (COUNT) { d_inner_it++; d_inner_act += (uint32_t)ni; }
                if (inner && QUANT == 2) {
                    // one 128-byte node = one L2 line: 16-bit plane coordinates of 8 children, near / far plane
                    // picked by the ray's signs; child 2k sits in the low half of dword k, child 2k+1 in the high half
                    const uint4 *nq = nodesq + 8 * (size_t)node;
                    const uint4 LX = nq[0], LY = nq[1], LZ = nq[2], HX = nq[3], HY = nq[4], HZ = nq[5], RA = nq[6], RB = nq[7];
                    const bool gx = nx != 0, gy = ny != 0, gz = nz != 0;
                    const uint4 NX = gx ? HX : LX, FX = gx ? LX : HX;
                    const uint4 NY = gy ? HY : LY, FY = gy ? LY : HY;
                    const uint4 NZ = gz ? HZ : LZ, FZ = gz ? LZ : HZ;
                    float k0, k1, k2, k3, k4, k5, k6, k7;
#define CRT_QBOX8(K, C, LOHI) { \
                        const float tn_ = __builtin_fmaxf(__builtin_fmaxf(fma_((float)(LOHI(NX.C)), id.x, oid.x), fma_((float)(LOHI(NY.C)), id.y, oid.y)), \
                                                          __builtin_fmaxf(fma_((float)(LOHI(NZ.C)), id.z, oid.z), t_min)); \
                        const float tf_ = __builtin_fminf(__builtin_fminf(fma_((float)(LOHI(FX.C)), id.x, oid.x), fma_((float)(LOHI(FY.C)), id.y, oid.y)), \
                                                          __builtin_fminf(fma_((float)(LOHI(FZ.C)), id.z, oid.z), t_max)); \
                        K = (tn_ <= tf_ * 1.0000005f) ? tn_ : 3.0e38f; }
#define CRT_LO16(v) ((v) & 0xFFFFu)
#define CRT_HI16(v) ((v) >> 16)
                    CRT_QBOX8(k0, x, CRT_LO16) CRT_QBOX8(k1, x, CRT_HI16) CRT_QBOX8(k2, y, CRT_LO16) CRT_QBOX8(k3, y, CRT_HI16)
                    CRT_QBOX8(k4, z, CRT_LO16) CRT_QBOX8(k5, z, CRT_HI16) CRT_QBOX8(k6, w, CRT_LO16) CRT_QBOX8(k7, w, CRT_HI16)
#undef CRT_QBOX8
#undef CRT_LO16
#undef CRT_HI16
                    int r0 = (int)RA.x, r1 = (int)RA.y, r2 = (int)RA.z, r3 = (int)RA.w, r4 = (int)RB.x, r5 = (int)RB.y, r6 = (int)RB.z, r7 = (int)RB.w;
                    if (COUNT) c_nodes += 8;
                    // sort the eight (key, ref) pairs by entry distance: Batcher's 19 compare-exchanges
#define CRT_CAS(ka, ra, kb, rb) { const bool sw_ = kb < ka; const float tk_ = sw_ ? kb : ka; kb = sw_ ? ka : kb; ka = tk_; \
                                  const int tr_ = sw_ ? rb : ra; rb = sw_ ? ra : rb; ra = tr_; }
                    CRT_CAS(k0, r0, k1, r1) CRT_CAS(k2, r2, k3, r3) CRT_CAS(k4, r4, k5, r5) CRT_CAS(k6, r6, k7, r7)
                    CRT_CAS(k0, r0, k2, r2) CRT_CAS(k1, r1, k3, r3) CRT_CAS(k4, r4, k6, r6) CRT_CAS(k5, r5, k7, r7)
                    CRT_CAS(k1, r1, k2, r2) CRT_CAS(k5, r5, k6, r6)
                    CRT_CAS(k0, r0, k4, r4) CRT_CAS(k1, r1, k5, r5) CRT_CAS(k2, r2, k6, r6) CRT_CAS(k3, r3, k7, r7)
                    CRT_CAS(k2, r2, k4, r4) CRT_CAS(k3, r3, k5, r5)
                    CRT_CAS(k1, r1, k2, r2) CRT_CAS(k3, r3, k4, r4) CRT_CAS(k5, r5, k6, r6)
#undef CRT_CAS
#define CRT_PUSH(K, R) if (K < 3.0e38f) { if (sp < kWfStack) stk[sp * 64] = R; else ovf[(size_t)(sp - kWfStack) * ovl] = R; sp++; }
                    if (k0 < 3.0e38f) {
                        // descend into the nearest; the others wait on the stack, farthest pushed first
                        CRT_PUSH(k7, r7) CRT_PUSH(k6, r6) CRT_PUSH(k5, r5) CRT_PUSH(k4, r4) CRT_PUSH(k3, r3) CRT_PUSH(k2, r2) CRT_PUSH(k1, r1)
                        node = r0;
                    } else if (sp > 0) {
                        sp--; node = stack_pop(stk, ovf, ovl, sp);
                    } else {
                        node = kNoNode;                                  // nothing left to walk ...
                        if (pend == 0) {                                 // ... and no postponed leaf either: the ray is through
                            if (shadow) stnt(&g_vis[slot], (b_slot == b_slot_in) ? 1u : 0u);
                            else stnt(&g_hit[slot], float2{t_max, bits_f(b_slot)});
                            active = false;
                        }
                    }
#undef CRT_PUSH
                } else if (inner) {
#if CRT_WF_BVH4
                  float k0, k1, k2, k3;
                  int r0, r1, r2, r3;
                  if (QUANT) {
                    // one 64-byte node: 16-bit plane coordinates, near / far plane picked by the ray's signs
                    const uint4 *nq = nodesq + 4 * (size_t)node;
                    const uint4 Q0 = nq[0], Q1 = nq[1], Q2 = nq[2], Q3 = nq[3];
                    const bool gx = nx != 0, gy = ny != 0, gz = nz != 0;
                    const uint32_t nxa = gx ? Q1.z : Q0.x, nxb = gx ? Q1.w : Q0.y, fxa = gx ? Q0.x : Q1.z, fxb = gx ? Q0.y : Q1.w;
                    const uint32_t nya = gy ? Q2.x : Q0.z, nyb = gy ? Q2.y : Q0.w, fya = gy ? Q0.z : Q2.x, fyb = gy ? Q0.w : Q2.y;
                    const uint32_t nza = gz ? Q2.z : Q1.x, nzb = gz ? Q2.w : Q1.y, fza = gz ? Q1.x : Q2.z, fzb = gz ? Q1.y : Q2.w;
#define CRT_QBOX(K, NXQ, NYQ, NZQ, FXQ, FYQ, FZQ) { \
                        const float tn_ = __builtin_fmaxf(__builtin_fmaxf(fma_((float)(NXQ), id.x, oid.x), fma_((float)(NYQ), id.y, oid.y)), \
                                                          __builtin_fmaxf(fma_((float)(NZQ), id.z, oid.z), t_min)); \
                        const float tf_ = __builtin_fminf(__builtin_fminf(fma_((float)(FXQ), id.x, oid.x), fma_((float)(FYQ), id.y, oid.y)), \
                                                          __builtin_fminf(fma_((float)(FZQ), id.z, oid.z), t_max)); \
                        K = (tn_ <= tf_ * 1.0000005f) ? tn_ : 3.0e38f; }
                    CRT_QBOX(k0, nxa & 0xFFFFu, nya & 0xFFFFu, nza & 0xFFFFu, fxa & 0xFFFFu, fya & 0xFFFFu, fza & 0xFFFFu)
                    CRT_QBOX(k1, nxa >> 16, nya >> 16, nza >> 16, fxa >> 16, fya >> 16, fza >> 16)
                    CRT_QBOX(k2, nxb & 0xFFFFu, nyb & 0xFFFFu, nzb & 0xFFFFu, fxb & 0xFFFFu, fyb & 0xFFFFu, fzb & 0xFFFFu)
                    CRT_QBOX(k3, nxb >> 16, nyb >> 16, nzb >> 16, fxb >> 16, fyb >> 16, fzb >> 16)
#undef CRT_QBOX
                    r0 = (int)Q3.x; r1 = (int)Q3.y; r2 = (int)Q3.z; r3 = (int)Q3.w;
                  } else {
                    // one 128-byte node: boxes of 4 children as SoA planes lo.x lo.y lo.z hi.x hi.y hi.z.
                    // The ray's direction signs pick the near / far plane per axis (no min/max per box).
                    const float4 *np = nodes + 8 * (size_t)node;
                    const float4 NX = np[nx], NY = np[1 + ny], NZ = np[2 + nz];
                    const float4 FX = np[3 - nx], FY = np[4 - ny], FZ = np[5 - nz];
                    const float4 RF = np[6];
                    {
                        float tn, tf;
                        tn = __builtin_fmaxf(__builtin_fmaxf(fma_(NX.x, id.x, -oid.x), fma_(NY.x, id.y, -oid.y)), __builtin_fmaxf(fma_(NZ.x, id.z, -oid.z), t_min));
                        tf = __builtin_fminf(__builtin_fminf(fma_(FX.x, id.x, -oid.x), fma_(FY.x, id.y, -oid.y)), __builtin_fminf(fma_(FZ.x, id.z, -oid.z), t_max));
                        k0 = (tn <= tf * 1.0000005f) ? tn : 3.0e38f;
                        tn = __builtin_fmaxf(__builtin_fmaxf(fma_(NX.y, id.x, -oid.x), fma_(NY.y, id.y, -oid.y)), __builtin_fmaxf(fma_(NZ.y, id.z, -oid.z), t_min));
                        tf = __builtin_fminf(__builtin_fminf(fma_(FX.y, id.x, -oid.x), fma_(FY.y, id.y, -oid.y)), __builtin_fminf(fma_(FZ.y, id.z, -oid.z), t_max));
                        k1 = (tn <= tf * 1.0000005f) ? tn : 3.0e38f;
                        tn = __builtin_fmaxf(__builtin_fmaxf(fma_(NX.z, id.x, -oid.x), fma_(NY.z, id.y, -oid.y)), __builtin_fmaxf(fma_(NZ.z, id.z, -oid.z), t_min));
                        tf = __builtin_fminf(__builtin_fminf(fma_(FX.z, id.x, -oid.x), fma_(FY.z, id.y, -oid.y)), __builtin_fminf(fma_(FZ.z, id.z, -oid.z), t_max));
                        k2 = (tn <= tf * 1.0000005f) ? tn : 3.0e38f;
                        tn = __builtin_fmaxf(__builtin_fmaxf(fma_(NX.w, id.x, -oid.x), fma_(NY.w, id.y, -oid.y)), __builtin_fmaxf(fma_(NZ.w, id.z, -oid.z), t_min));
                        tf = __builtin_fminf(__builtin_fminf(fma_(FX.w, id.x, -oid.x), fma_(FY.w, id.y, -oid.y)), __builtin_fminf(fma_(FZ.w, id.z, -oid.z), t_max));
                        k3 = (tn <= tf * 1.0000005f) ? tn : 3.0e38f;
                    }
                    r0 = (int)f_bits(RF.x); r1 = (int)f_bits(RF.y); r2 = (int)f_bits(RF.z); r3 = (int)f_bits(RF.w);
                  }
                    if (COUNT) c_nodes += 4;
                    // sort the four (key, ref) pairs by entry distance: 5 compare-exchanges
#define CRT_CAS(ka, ra, kb, rb) { const bool sw_ = kb < ka; const float tk_ = sw_ ? kb : ka; kb = sw_ ? ka : kb; ka = tk_; \
                                  const int tr_ = sw_ ? rb : ra; rb = sw_ ? ra : rb; ra = tr_; }
                    CRT_CAS(k0, r0, k1, r1) CRT_CAS(k2, r2, k3, r3) CRT_CAS(k0, r0, k2, r2) CRT_CAS(k1, r1, k3, r3) CRT_CAS(k1, r1, k2, r2)
#undef CRT_CAS
                    if (k0 < 3.0e38f) {
                        // descend into the nearest; the others wait on the stack, farthest pushed first
                        if (k3 < 3.0e38f) { if (sp < kWfStack) stk[sp * 64] = r3; else ovf[(size_t)(sp - kWfStack) * ovl] = r3; sp++; }
                        if (k2 < 3.0e38f) { if (sp < kWfStack) stk[sp * 64] = r2; else ovf[(size_t)(sp - kWfStack) * ovl] = r2; sp++; }
                        if (k1 < 3.0e38f) { if (sp < kWfStack) stk[sp * 64] = r1; else ovf[(size_t)(sp - kWfStack) * ovl] = r1; sp++; }
                        node = r0;
                    } else if (sp > 0) {
                        sp--; node = stack_pop(stk, ovf, ovl, sp);
                    } else {
                        node = kNoNode;                                  // nothing left to walk ...
                        if (pend == 0) {                                 // ... and no postponed leaf either: the ray is through
                            if (shadow) stnt(&g_vis[slot], (b_slot == b_slot_in) ? 1u : 0u);
                            else stnt(&g_hit[slot], float2{t_max, bits_f(b_slot)});
                            active = false;
                        }
                    }
#else
                    const float4 *np = nodes + 4 * (size_t)node;
                    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
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
                    const bool first0 = tn0 <= tn1;
                    if (h0 & h1) {
                        stk[sp * 64] = first0 ? r1 : r0;
                        sp++;
                        node = first0 ? r0 : r1;
                    } else if (h0 | h1) {
                        node = h0 ? r0 : r1;
                    } else if (sp > 0) {
                        sp--; node = stk[sp * 64];
                    } else {
                        node = kNoNode;                                  // stack empty ...
                        if (pend == 0) {                                 // ... and no postponed leaf: this ray is finished
                            if (shadow) stnt(&g_vis[slot], (b_slot == b_slot_in) ? 1u : 0u);
                            else stnt(&g_hit[slot], float2{t_max, bits_f(b_slot)});
                            active = false;
                        }
                    }
#endif
                }
            }
        }
    }
    if (COUNT) {
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_NODES, c_nodes);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_PRIMS, c_prims);
        if (lane == 0) {
            atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 8, (unsigned long long)d_inner_it); atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 9, (unsigned long long)d_inner_act);
            atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 10, (unsigned long long)d_leaf_it); atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 11, (unsigned long long)d_leaf_act);
            atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 12, (unsigned long long)d_prim_it); atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 13, (unsigned long long)d_scans); atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 14, (unsigned long long)d_refill);
            atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 15, (unsigned long long)d_refill_lanes);
        }
    }
}

// ------------------------------------------------------------------ trace, second form
// The same walk with the lanes regrouped by what they do next (round-2 verdict, item 1):
//  * RAY RING.  A chunk of up to 64 rays is fetched and SET UP by the whole wave at once (record loads, the three
//    reciprocals, the folded plane constants: every lane on the same code) into a ring of ready-to-run rays in LDS, and
//    idle lanes pick rays up from it whenever CRT_WF_PICK_AT of them are idle: a pick-up is four DS reads, so it can
//    run at a granularity at which the first form's refill (global loads + set-up under a quarter-filled mask) could
//    not -- the inner-node passes no longer carry a dozen finished lanes on average.
//  * PRIMITIVE TASKS.  A leaf pass does not loop over each lane's own leaf (lanes with one triangle idling behind the
//    lane with four): the waiting leaves are expanded into (ray, primitive) tasks in LDS, compacted over the wave with
//    bit-plane prefix counts, and lane k runs task k -- the ray comes over from its owner's registers (ds_bpermute),
//    the result goes back through an LDS cell per ray, `min` over the 64-bit key (t bits, 0xFFFFFFFE - index): the
//    order-independent form of the reference's rule "closest t, equal t -> later primitive" (:557,:609), so any number
//    of tasks of one ray may run side by side.  One pass = at most 64 tasks, started once CRT_WF_LEAF2_AT lanes hold a leaf.
// Stack: 16 entries per lane in LDS, the rest in the global overflow area.  LDS: 4 + 2.25 + 1 KB per wave.
#ifndef CRT_WF_PICK_AT
#define CRT_WF_PICK_AT 8
#endif
#ifndef CRT_WF_PKFMA
#define CRT_WF_PKFMA 0
#endif
#ifndef CRT_WF_LEAF2_AT
#define CRT_WF_LEAF2_AT 28
#endif
#ifndef CRT_WF_STALL_AT
#define CRT_WF_STALL_AT 16
#endif
constexpr int kStk2 = 16;
#ifndef CRT_WF_RING
#define CRT_WF_RING 32
#endif
constexpr uint32_t kRing2 = CRT_WF_RING;        // ready rays per refill of the ring (LDS: 7.3 KB per wave with 32, so that a few waves of the
                                                // other pipe's traversal launch fit beside sixteen of this one and the two launches' tail and ramp-up overlap)

__device__ __forceinline__ int stack_pop2(const int *stk, const int *ovf, size_t ovl, int sp)
{
    int v = stk[(sp < kStk2 ? sp : kStk2 - 1) * 64];
    if (sp >= kStk2) v = *(const volatile int *)(ovf + (size_t)(sp - kStk2) * ovl);   // (volatile: keeps it apart from the DS read)
    return v;
}
// lanes below this one whose bit is set in m
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// (t, index) of a hit as one ordered key: smaller = better under the reference's rule; index none -> low word all ones
__device__ __forceinline__ unsigned long long hit_key(float t, uint32_t index)
{
    return ((unsigned long long)f_bits(t) << 32) | (unsigned long long)(index == kNoHit ? 0xFFFFFFFFu : 0xFFFFFFFEu - index);
}

#ifndef CRT_WF_T2_WAVES
#define CRT_WF_T2_WAVES 5
#endif
template <bool COUNT>
__global__ __launch_bounds__(64, CRT_WF_T2_WAVES) void k_wf_trace2(const WfParams P, uint32_t it)
{
    __shared__ int lds_stack[kStk2 * 64];
    __shared__ float4 ring[4][kRing2];           // ready rays: (o, exclude) (d, slot | shadow << 31) (id', t_max) (oid', -)
    __shared__ uint2 ring_l[kRing2];             //             (b_index, b_slot): the light's primitive for a shadow ray
    __shared__ unsigned long long cell[64];      // per lane: hit_key of its ray's best hit so far
    __shared__ uint32_t cslot[64];               // ... and that primitive's slot
    __shared__ uint2 tasks[64];                  // (primitive slot, owner lane): written by the owner, so a task lane's loads start after ONE DS read
    WfCtl *ctl = P.ctl;
    const uint4 *__restrict__ nodesq = P.sc.nodes4q;
    const f3 qscale = f3{P.sc.qscale[0], P.sc.qscale[1], P.sc.qscale[2]}, qbase = f3{P.sc.qbase[0], P.sc.qbase[1], P.sc.qbase[2]};
    const float4 *__restrict__ prim = P.sc.prim;
    const float4 *__restrict__ primD = P.sc.primD;
    uint32_t *__restrict__ g_vis = P.vis;
    float2 *__restrict__ g_hit = P.hit;
    const size_t cls_stride = (size_t)P.list_cap * kWfShards;
    const float4 *__restrict__ recA = P.recA + (size_t)((it & 1u) * 4u) * cls_stride;
    const float4 *__restrict__ recB = P.recB + (size_t)((it & 1u) * 4u) * cls_stride;
    const uint4 *__restrict__ recC = P.recC + (size_t)((it & 1u) * 4u) * cls_stride;
    const float hit_pad = P.sc.hit_pad;
    const int root = P.sc.root4;
    int *__restrict__ ovf = P.stack_overflow + ((size_t)blockIdx.x * 64 + lane_id());
    const size_t ovl = P.overflow_lanes;
    const uint32_t nprim = P.sc.nprim;
    const bool fetchD = ((unsigned long long)P.sc.npatch << CRT_WF_PATCH_RECOMPUTE) > (unsigned long long)nprim;
    DevScene S = P.sc;                                         // for the rare patch / sphere tests
    S.prim = prim; S.primD = primD;

    const uint32_t ringi = it & 3u;
    const uint32_t lane = lane_id();
    int *stk = lds_stack + lane;
    const float t_min = 0.001f;

    // wave-uniform fetch state: current shard, its class end offsets, the reserved chunk [pos,end), the ring [head, head+n)
    uint32_t cur_shard = blockIdx.x % kWfShards, sh_e0 = 0, sh_e1 = 0, sh_e2 = 0, sh_total = 0;
    uint32_t chunk_pos = 0, chunk_end = 0, ring_head = 0, ring_n = 0;
    bool have_shard = false, exhausted = false;
    {
        const WfShard &so = ctl->shard[ringi][cur_shard];
        sh_e0 = so.n[0]; sh_e1 = sh_e0 + so.n[1]; sh_e2 = sh_e1 + so.n[2]; sh_total = sh_e2 + so.n[3];
        have_shard = sh_total > 0u;
    }
    bool active = false;
    int pend = 0;                                   // a leaf put aside, 0 = none
    f3 o = f3{0, 0, 0}, d = f3{0, 0, 0}, id = f3{0, 0, 0}, oid = f3{0, 0, 0};
    uint32_t excl = 0, slot = 0, b_index = kNoHit, b_slot = kNoHit, b_slot_in = kNoHit;
    float t_max = 0.0f;
    bool shadow = false;
    int node = kNoNode, sp = 0;
    bool gx = false, gy = false, gz = false;         // the hi plane is the near one on that axis
    uint32_t c_nodes = 0, c_prims = 0;
    uint32_t d_inner_it = 0, d_inner_act = 0, d_leaf_it = 0, d_leaf_act = 0, d_prim_it = 0, d_refill = 0, d_refill_lanes = 0, d_scans = 0;   // lane 0 only

    for (;;) {
        // ---- idle lanes pick ready rays up from the ring; an empty ring is refilled with the next chunk first
        const unsigned long long idle = __ballot(!active);
        const int nidle = __popcll(idle);
        if (nidle >= CRT_WF_PICK_AT) {
            if (ring_n == 0u && !exhausted) {
                for (int guard = 0; chunk_pos == chunk_end && !exhausted && guard < 2 * (int)kWfShards; guard++) {
                    if (have_shard) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&ctl->shard[ringi][cur_shard].cur, (uint32_t)kTraceChunk);
                        base = __shfl(base, 0, 64);
                        if (base < sh_total) { chunk_pos = base; chunk_end = min(base + (uint32_t)kTraceChunk, sh_total); break; }
                        have_shard = false;
                    }
                    if (COUNT) d_scans++;
                    const WfShard &sl = ctl->shard[ringi][lane % kWfShards];
                    const uint32_t e0 = sl.n[0], e1 = e0 + sl.n[1], e2 = e1 + sl.n[2], tot = e2 + sl.n[3];
                    const uint32_t cur_l = __hip_atomic_load(&sl.cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long avail = __ballot(cur_l < tot);
                    if (!avail) { exhausted = true; break; }
                    const uint32_t rot = cur_shard & 63u;
                    const unsigned long long rmask = rot ? ((avail >> rot) | (avail << (64u - rot))) : avail;
                    cur_shard = ((uint32_t)(__ffsll((long long)rmask) - 1) + rot) & 63u;
                    sh_e0 = __shfl(e0, (int)cur_shard, 64); sh_e1 = __shfl(e1, (int)cur_shard, 64);
                    sh_e2 = __shfl(e2, (int)cur_shard, 64); sh_total = __shfl(tot, (int)cur_shard, 64);
                    have_shard = true;
                }
                if (chunk_pos < chunk_end) {
                    // the next (up to) kRing2 rays of the chunk, set up by as many lanes at once
                    const uint32_t n = min(kRing2, chunk_end - chunk_pos);
                    bool ok = false;
                    float4 R0 = float4{0, 0, 0, 0}, R1 = R0, R2 = R0, R3 = R0;
                    uint2 RL = uint2{kNoHit, kNoHit};
                    if (lane < n) {
                        const uint32_t idx = chunk_pos + lane;
                        const size_t region = (size_t)cur_shard * P.list_cap;
                        const bool sh = idx >= sh_e1;
                        const size_t g = idx < sh_e0 ? region + idx : idx < sh_e1 ? cls_stride + region + (idx - sh_e0)
                                       : idx < sh_e2 ? 2u * cls_stride + region + (idx - sh_e1) : 3u * cls_stride + region + (idx - sh_e2);
                        const float4 ro = ldnt(&recA[g]), rd = ldnt(&recB[g]);
                        uint4 rc = uint4{0, 0, 0, 0};
                        if (sh) rc = ldnt(&recC[g]);
                        uint32_t r_slot;
                        float r_tmax;
                        ok = true;
                        if (sh) {
                            r_slot = rc.x & kWfListSlot; r_tmax = rd.w;   // (the mark on rc.x is tail mode's business)
                            RL = uint2{rc.y, rc.z};                      // the light's primitive index and slot
                        } else {
                            r_slot = f_bits(rd.w) & kWfListSlot; r_tmax = CRT_INFINITY;
                            if (f_bits(rd.w) & kWfListAlsoExt) ok = false;   // non-finite ray, already resolved by k_wf_shade
                            else if (nprim == 0u) { g_hit[r_slot] = float2{r_tmax, bits_f(kNoHit)}; ok = false; }
                        }
                        const float tiny = 1.0e-20f;
                        f3 i3;
                        i3.x = 1.0f / (abs_(rd.x) > tiny ? rd.x : __builtin_copysignf(tiny, rd.x));
                        i3.y = 1.0f / (abs_(rd.y) > tiny ? rd.y : __builtin_copysignf(tiny, rd.y));
                        i3.z = 1.0f / (abs_(rd.z) > tiny ? rd.z : __builtin_copysignf(tiny, rd.z));
                        // plane = qbase + q*qscale  =>  t = q*(qscale*id) + (qbase*id - o*id): one fma per plane
                        const f3 oi = f3{fma_(qbase.x, i3.x, -(ro.x * i3.x)), fma_(qbase.y, i3.y, -(ro.y * i3.y)), fma_(qbase.z, i3.z, -(ro.z * i3.z))};
                        R0 = ro;
                        R1 = float4{rd.x, rd.y, rd.z, bits_f(r_slot | (sh ? 0x80000000u : 0u))};
                        // (the sign of the raw reciprocal picks the near planes: kept in the w of R3, the scaled one may be +-0)
                        R2 = float4{qscale.x * i3.x, qscale.y * i3.y, qscale.z * i3.z, r_tmax};
                        R3 = float4{oi.x, oi.y, oi.z, bits_f((i3.x < 0.0f ? 1u : 0u) | (i3.y < 0.0f ? 2u : 0u) | (i3.z < 0.0f ? 4u : 0u))};
                    }
                    const unsigned long long mk = __ballot(ok);
                    if (ok) {
                        const uint32_t e = mbcnt64(mk);
                        ring[0][e] = R0; ring[1][e] = R1; ring[2][e] = R2; ring[3][e] = R3; ring_l[e] = RL;
                    }
                    ring_head = 0; ring_n = (uint32_t)__popcll(mk);
                    chunk_pos += n;
                    if (COUNT) { d_refill++; d_refill_lanes += ring_n; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (ring_n > 0u) {
                const uint32_t give = min((uint32_t)nidle, ring_n);
                const uint32_t my = mbcnt64(idle);
                if (!active && my < give) {
                    const uint32_t e = ring_head + my;                   // (the ring is refilled from 0, only when empty: no wrap)
                    const float4 R0 = ring[0][e], R1 = ring[1][e], R2 = ring[2][e], R3 = ring[3][e];
                    const uint2 RL = ring_l[e];
                    o = xyz(R0); excl = f_bits(R0.w);
                    d = xyz(R1); slot = f_bits(R1.w) & kWfListSlot; shadow = (f_bits(R1.w) >> 31) != 0u;
                    id = xyz(R2); t_max = R2.w;
                    oid = xyz(R3);
                    const uint32_t sg = f_bits(R3.w);
                    gx = (sg & 1u) != 0u; gy = (sg & 2u) != 0u; gz = (sg & 4u) != 0u;
                    b_index = RL.x; b_slot = RL.y; b_slot_in = b_slot;
                    cell[lane] = hit_key(t_max, b_index); cslot[lane] = b_slot;
                    node = root; sp = 0; pend = 0;
                    active = true;
                }
                ring_head += give; ring_n -= give;
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted && ring_n == 0u) break;
            continue;
        }

#pragma unroll 1
        for (int pass = 0; pass < 64; pass++) {
            // A lane that reaches a leaf puts it aside (one per lane) and goes on with the next node of its stack: the
            // closest hit does not depend on the order, only t_max shrinks a little later.
            if (active && node < 0 && pend == 0 && sp > 0) {
                pend = node;
                sp--; node = stack_pop2(stk, ovf, ovl, sp);
            }
            const bool inner = active && node >= 0 && node != kNoNode;
            const bool leaf = active && (node < 0 || pend != 0);
            // (an active lane is on an inner node, or holds a leaf, or both: nb - ni lanes cannot take an inner-node step)
            const int ni = __popcll(__ballot(inner)), nl = __popcll(__ballot(leaf)), nb = __popcll(__ballot(active));
            if (nb == 0) break;
            if (pass > 0 && 64 - nb >= CRT_WF_PICK_AT && (ring_n > 0u || !exhausted)) break;   // enough idle lanes: pick up first
            if (nl >= CRT_WF_LEAF2_AT || ni == 0 || nb - ni >= CRT_WF_STALL_AT) {
                // ---- leaf pass: the waiting leaves' primitives as one compacted round of tasks
                const int lf = pend != 0 ? pend : node;                  // the leaf this lane offers (the postponed one first)
                const uint32_t cnt = leaf ? ((~(uint32_t)lf) & 7u) + 1u : 0u;
                const unsigned long long c0 = __ballot((cnt & 1u) != 0u), c1 = __ballot((cnt & 2u) != 0u), c2 = __ballot((cnt & 4u) != 0u), c3 = __ballot((cnt & 8u) != 0u);
                const uint32_t pre = mbcnt64(c0) + 2u * mbcnt64(c1) + 4u * mbcnt64(c2) + 8u * mbcnt64(c3);
                const bool incl = leaf && pre + cnt <= 64u;              // (whole leaves only; the included lanes are a prefix of the leaf lanes)
                const unsigned long long mi = __ballot(incl);
                const uint32_t T = (uint32_t)(__popcll(c0 & mi) + 2 * __popcll(c1 & mi) + 4 * __popcll(c2 & mi) + 8 * __popcll(c3 & mi));
                const uint32_t first = (~(uint32_t)lf) >> 3;
                if (incl) {                                              // (the builders' leaves hold one to four primitives as a rule)
                    tasks[pre] = uint2{first, lane};
                    if (cnt > 1u) tasks[pre + 1u] = uint2{first + 1u, lane};
                    if (cnt > 2u) tasks[pre + 2u] = uint2{first + 2u, lane};
                    if (cnt > 3u) tasks[pre + 3u] = uint2{first + 3u, lane};
                    for (uint32_t i = 4; i < cnt; i++) tasks[pre + i] = uint2{first + i, lane};
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (COUNT) { d_leaf_it++; d_leaf_act += T; d_prim_it += 1u; }
                // every lane shuffles (a bpermute reads active lanes only); lanes beyond T run on their own ray and do nothing
                const uint2 tk = lane < T ? tasks[lane] : uint2{0u, lane};
                const int owner = (int)tk.y;
                const uint32_t ps = tk.x;
                // (the record loads go out first: the ray's values come over under their latency)
                float4 A = float4{0, 0, 0, 0}, B = A, C = A, D = A;
                // (The fourth part of a record is a patch's: unit normal and e1.e1.  Where a patch task is in nearly every round -- S2's
                // walls: 94 % -- it is fetched with the other three (behind the category test it was a dependent round trip per
                // round).  Where patches are a handful among millions of triangles -- the 10 M soup -- that fetch is one memory
                // request in four of EVERY task for nothing, and the kernel is bound by its requests: left out, the rare patch task
                // recomputes the part below, +5 % Mrays/s.  Recomputing always is +-0 in time on S2 but +8 % VALU instructions at
                // 61 instead of 64 % lane utilisation: profiles/r03_ab_patch_recompute.txt.)
                if (lane < T) { A = prim[3 * (size_t)ps + 0]; B = prim[3 * (size_t)ps + 1]; C = prim[3 * (size_t)ps + 2]; if (fetchD) D = primD[ps]; }
                // the owners' next node meanwhile (it does not depend on the tests' outcome)
                bool from_pend = false;
                if (incl) {
                    from_pend = pend != 0;
                    if (from_pend) pend = 0;                             // the postponed leaf went first
                    else if (sp > 0) { sp--; node = stack_pop2(stk, ovf, ovl, sp); }
                    else node = kNoNode;
                }
                const f3 to = f3{__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64)};
                const f3 td = f3{__shfl(d.x, owner, 64), __shfl(d.y, owner, 64), __shfl(d.z, owner, 64)};
                const uint32_t t_excl = (uint32_t)__shfl((int)excl, owner, 64);
                bool acc = false;
                unsigned long long key = 0;
                if (lane < T) {
                    const unsigned long long cur = cell[owner];
                    float tm = bits_f((uint32_t)(cur >> 32));
                    const uint32_t low = (uint32_t)cur;
                    uint32_t bi = low == 0xFFFFFFFFu ? kNoHit : 0xFFFFFFFEu - low;
                    const uint32_t bs0 = low == 0xFFFFFFFFu ? kNoHit : 0xFFFFFFFEu;     // "some hit" (no primitive has that slot)
                    uint32_t bs = bs0;
                    // (D, the patch record's fourth part, was fetched with the other three: 94 % of the rounds on S2 hold a wall patch
                    // among their tasks, and fetched behind the category test it was a dependent round trip in every one of them)
                    if ((f_bits(A.w) & 3u) == 2u) tri_test(A, B, C, ps, to, td, t_excl, t_min, hit_pad, tm, bi, bs);
                    else {
                        // (unit normal and e1.e1 with the operations of the upload, crt_api.cpp: the same bits as the stored part)
                        if (!fetchD && (f_bits(A.w) & 3u) == 0u) { const f3 nn = normalize(cross(xyz(B), xyz(C))); D = float4{nn.x, nn.y, nn.z, dot(xyz(B), xyz(B))}; }
                        hit_test_rec<false>(A, B, C, D, hit_pad, ps, to, td, t_excl, t_min, tm, bi, bs);
                    }
                    if (COUNT && __ballot((f_bits(A.w) & 3u) != 2u) != 0ull) d_scans += 64u;   // (probe: rounds that also run the patch / sphere test; read back / 64)
                    acc = bs != bs0;
                    if (acc) {
                        key = hit_key(tm, bi);
                        atomicMin(&cell[owner], key);
                    }
                    if (COUNT) c_prims++;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (acc && cell[owner] == key) cslot[owner] = ps;        // the round's winner for that ray names its slot
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (incl) {
                    const unsigned long long v = cell[lane];
                    const uint32_t low = (uint32_t)v;
                    t_max = bits_f((uint32_t)(v >> 32));
                    b_index = low == 0xFFFFFFFFu ? kNoHit : 0xFFFFFFFEu - low;
                    b_slot = cslot[lane];
                    bool done = false;
                    if (shadow && b_slot != b_slot_in) done = true;      // any-hit: something beats the light
                    else if (node == kNoNode && pend == 0) done = true;
                    if (done) {
                        if (shadow) stnt(&g_vis[slot], (b_slot == b_slot_in) ? 1u : 0u);
                        else stnt(&g_hit[slot], float2{t_max, bits_f(b_slot)});
                        active = false;
                    }
                }
            } else {
                if (COUNT) { d_inner_it++; d_inner_act += (uint32_t)ni; }
                if (inner) {
                    float k0, k1, k2, k3;
                    int r0, r1, r2, r3;
                    // one 64-byte node: 16-bit plane coordinates, near / far plane picked by the ray's signs
                    const uint4 *nq = (const uint4 *)((const char *)nodesq + ((uint32_t)node << 6));   // (32-bit offset: the tree is below 4 GB)
                    const uint4 Q0 = nq[0], Q1 = nq[1], Q2 = nq[2], Q3 = nq[3];
                    const uint32_t nxa = gx ? Q1.z : Q0.x, nxb = gx ? Q1.w : Q0.y, fxa = gx ? Q0.x : Q1.z, fxb = gx ? Q0.y : Q1.w;
                    const uint32_t nya = gy ? Q2.x : Q0.z, nyb = gy ? Q2.y : Q0.w, fya = gy ? Q0.z : Q2.x, fyb = gy ? Q0.w : Q2.y;
                    const uint32_t nza = gz ? Q2.z : Q1.x, nzb = gz ? Q2.w : Q1.y, fza = gz ? Q1.x : Q2.z, fzb = gz ? Q1.y : Q2.w;
#if CRT_WF_PKFMA
                    // the 24 plane distances as 12 packed fmas (v_pk_fma_f32: children 0/1 and 2/3 of a plane share a dword).  Measured 4 %
                    // SLOWER on the step (profiles/r03_ab_pkfma.txt: the packed form pairs registers and spills in the set-up): off.
                    {
                        const v2f ix = v2f{id.x, id.x}, iy = v2f{id.y, id.y}, iz = v2f{id.z, id.z};
                        const v2f ox = v2f{oid.x, oid.x}, oy = v2f{oid.y, oid.y}, oz = v2f{oid.z, oid.z};
#define CRT_PK(Q, I, O) __builtin_elementwise_fma(v2f{(float)((Q) & 0xFFFFu), (float)((Q) >> 16)}, I, O)
                        const v2f tnx_a = CRT_PK(nxa, ix, ox), tnx_b = CRT_PK(nxb, ix, ox), tfx_a = CRT_PK(fxa, ix, ox), tfx_b = CRT_PK(fxb, ix, ox);
                        const v2f tny_a = CRT_PK(nya, iy, oy), tny_b = CRT_PK(nyb, iy, oy), tfy_a = CRT_PK(fya, iy, oy), tfy_b = CRT_PK(fyb, iy, oy);
                        const v2f tnz_a = CRT_PK(nza, iz, oz), tnz_b = CRT_PK(nzb, iz, oz), tfz_a = CRT_PK(fza, iz, oz), tfz_b = CRT_PK(fzb, iz, oz);
#undef CRT_PK
#define CRT_QK(K, NX, NY, NZ, FX, FY, FZ) { \
                            const float tn_ = __builtin_fmaxf(__builtin_fmaxf(NX, NY), __builtin_fmaxf(NZ, t_min)); \
                            const float tf_ = __builtin_fminf(__builtin_fminf(FX, FY), __builtin_fminf(FZ, t_max)); \
                            K = (tn_ <= tf_ * 1.0000005f) ? tn_ : 3.0e38f; }
                        CRT_QK(k0, tnx_a.x, tny_a.x, tnz_a.x, tfx_a.x, tfy_a.x, tfz_a.x)
                        CRT_QK(k1, tnx_a.y, tny_a.y, tnz_a.y, tfx_a.y, tfy_a.y, tfz_a.y)
                        CRT_QK(k2, tnx_b.x, tny_b.x, tnz_b.x, tfx_b.x, tfy_b.x, tfz_b.x)
                        CRT_QK(k3, tnx_b.y, tny_b.y, tnz_b.y, tfx_b.y, tfy_b.y, tfz_b.y)
#undef CRT_QK
                    }
#else
#define CRT_QBOX(K, NXQ, NYQ, NZQ, FXQ, FYQ, FZQ) { \
                        const float tn_ = __builtin_fmaxf(__builtin_fmaxf(fma_((float)(NXQ), id.x, oid.x), fma_((float)(NYQ), id.y, oid.y)), \
                                                          __builtin_fmaxf(fma_((float)(NZQ), id.z, oid.z), t_min)); \
                        const float tf_ = __builtin_fminf(__builtin_fminf(fma_((float)(FXQ), id.x, oid.x), fma_((float)(FYQ), id.y, oid.y)), \
                                                          __builtin_fminf(fma_((float)(FZQ), id.z, oid.z), t_max)); \
                        K = (tn_ <= tf_ * 1.0000005f) ? tn_ : 3.0e38f; }
                    CRT_QBOX(k0, nxa & 0xFFFFu, nya & 0xFFFFu, nza & 0xFFFFu, fxa & 0xFFFFu, fya & 0xFFFFu, fza & 0xFFFFu)
                    CRT_QBOX(k1, nxa >> 16, nya >> 16, nza >> 16, fxa >> 16, fya >> 16, fza >> 16)
                    CRT_QBOX(k2, nxb & 0xFFFFu, nyb & 0xFFFFu, nzb & 0xFFFFu, fxb & 0xFFFFu, fyb & 0xFFFFu, fzb & 0xFFFFu)
                    CRT_QBOX(k3, nxb >> 16, nyb >> 16, nzb >> 16, fxb >> 16, fyb >> 16, fzb >> 16)
#undef CRT_QBOX
#endif
                    r0 = (int)Q3.x; r1 = (int)Q3.y; r2 = (int)Q3.z; r3 = (int)Q3.w;
                    if (COUNT) c_nodes += 4;
                    // sort the four (key, ref) pairs by entry distance: 5 compare-exchanges
#define CRT_CAS(ka, ra, kb, rb) { const bool sw_ = kb < ka; const float tk_ = sw_ ? kb : ka; kb = sw_ ? ka : kb; ka = tk_; \
                                  const int tr_ = sw_ ? rb : ra; rb = sw_ ? ra : rb; ra = tr_; }
                    CRT_CAS(k0, r0, k1, r1) CRT_CAS(k2, r2, k3, r3) CRT_CAS(k0, r0, k2, r2) CRT_CAS(k1, r1, k3, r3) CRT_CAS(k1, r1, k2, r2)
#undef CRT_CAS
                    if (k0 < 3.0e38f) {
                        // descend into the nearest; the others wait on the stack, farthest pushed first
                        if (k3 < 3.0e38f) { if (sp < kStk2) stk[sp * 64] = r3; else ovf[(size_t)(sp - kStk2) * ovl] = r3; sp++; }
                        if (k2 < 3.0e38f) { if (sp < kStk2) stk[sp * 64] = r2; else ovf[(size_t)(sp - kStk2) * ovl] = r2; sp++; }
                        if (k1 < 3.0e38f) { if (sp < kStk2) stk[sp * 64] = r1; else ovf[(size_t)(sp - kStk2) * ovl] = r1; sp++; }
                        node = r0;
                    } else if (sp > 0) {
                        sp--; node = stack_pop2(stk, ovf, ovl, sp);
                    } else {
                        node = kNoNode;                                  // nothing left to walk ...
                        if (pend == 0) {                                 // ... and no postponed leaf either: the ray is through
                            if (shadow) stnt(&g_vis[slot], (b_slot == b_slot_in) ? 1u : 0u);
                            else stnt(&g_hit[slot], float2{t_max, bits_f(b_slot)});
                            active = false;
                        }
                    }
                }
            }
        }
    }
    if (COUNT) {
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_NODES, c_nodes);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_PRIMS, c_prims);
        if (lane == 0) {
            atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 8, (unsigned long long)d_inner_it); atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 9, (unsigned long long)d_inner_act);
            atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 10, (unsigned long long)d_leaf_it); atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 11, (unsigned long long)d_leaf_act);
            atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 12, (unsigned long long)d_prim_it); atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 13, (unsigned long long)d_scans); atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 14, (unsigned long long)d_refill);
            atomicAdd(ctl->counters[blockIdx.x % kWfShards] + 15, (unsigned long long)d_refill_lanes);
        }
    }
}

// ------------------------------------------------------------------ finish
// The last few paths of a batch (no work left, a few thousand slots alive at most): one lane
// per remaining slot runs its path to the end -- trace the pending rays itself (plain BVH2 walk,
// any-hit for the shadow ray), shade, repeat -- so the tail costs one ray latency per bounce
// instead of two kernel launches per bounce.  Slots come from the previous iteration's ray lists
// exactly as in k_wf_shade's tail mode.
// ------------------------------------------------------------------ stragglers
// k_wf_finish: one lane per side-pool path, run to the path's end (shade step, then trace what it emits,
// ...) with the single-ray BVH2 walk.  Runs on its own stream while the pool works on the next batch.
template <bool COUNT>
__global__ __launch_bounds__(64) void k_wf_finish(const WfParams P, const WfFinishSegs G)
{
    __shared__ int lds_stack[kStackDepth * 64];
    const DevScene &S = P.sc;
    const uint32_t seg = blockIdx.x / G.blocks_per_seg, blk = blockIdx.x % G.blocks_per_seg;
    WfCtl *ctl = G.ctl[seg];
    int *stk = lds_stack + lane_id();
    // P.tail_bound = paths per wave: a wave runs until its longest path ends and every bounce costs the
    // slowest lane's walk, so when the GPU has nothing else to do few paths per wave finish sooner
    const uint32_t j = blk * P.tail_bound + threadIdx.x;
    const uint32_t count = min(ctl->side_count[G.batch[seg]], kWfSideCap);
    // (the host sizes the grid by the most paths a pipe reported for the batch; paths beyond it would never be
    // finished and resolve would read a stale staging cell: reported, and the host turns it into CRT_EDEVICE)
    if (blk == 0 && threadIdx.x == 0 && count > G.blocks_per_seg * P.tail_bound) atomicAdd(&ctl->dropped, count - G.blocks_per_seg * P.tail_bound);
    const bool mine = threadIdx.x < P.tail_bound && j < count;
    const uint32_t slot = G.base[seg] + (mine ? j : 0u);
    uint32_t flags = 0;
    if (mine) flags = P.misc[slot].y;
    bool alive = mine && (flags & kWfAlive);
    // The rays these paths listed in their last pool iteration were traced there: start with the shade
    // step; from then on this lane traces what its own shade steps emit.
    bool pend_sh = false, pend_ext = false;
    const bool wide = S.nodes4q != nullptr && S.nodes8q == nullptr;     // walk the quantised 4-wide tree when there is one
    ShadeCnt cn;
    uint32_t c_nodes = 0, c_prims = 0;
    for (int guard = 0; guard < 512 && __ballot(alive) != 0ull; guard++) {
        if (alive) {
            const float4 ro = P.ray_o[slot];
            const f3 o = xyz(ro);
            const uint32_t excl = f_bits(ro.w);
            if (pend_sh) {
                const float4 sd = P.sh_d[slot];
                float t_max = sd.w;
                uint32_t b_index = P.vis[slot];                          // the light's primitive index
                const uint32_t l_slot = S.slot_of_index[b_index];
                uint32_t b_slot = l_slot;
                if (!wide || !traverse4q<COUNT>(S, stk, o, xyz(sd), excl, true, t_max, b_index, b_slot, c_nodes, c_prims))
                    traverse<COUNT>(S, stk, o, xyz(sd), excl, true, t_max, b_index, b_slot, c_nodes, c_prims);
                P.vis[slot] = (b_slot == l_slot) ? 1u : 0u;
            }
            if (pend_ext) {
                const float4 rd = P.ray_d[slot];
                if (f_bits(rd.w) == 0u) {                                // (else: non-finite ray, resolved by the shade step)
                    float t_max = CRT_INFINITY;
                    uint32_t b_index = kNoHit, b_slot = kNoHit;
                    if (!wide || !traverse4q<COUNT>(S, stk, o, xyz(rd), excl, false, t_max, b_index, b_slot, c_nodes, c_prims))
                        traverse<COUNT>(S, stk, o, xyz(rd), excl, false, t_max, b_index, b_slot, c_nodes, c_prims);
                    P.hit[slot] = float2{t_max, bits_f(b_slot)};
                }
            }
            ShadeOut so = shade_body<COUNT, true>(P, slot, true, cn);
            const uint32_t resolved = shade_store<COUNT, true>(P, slot, true, so, cn);
            (void)resolved;                                              // (read back from ray_d.w above)
            alive = so.alive; pend_ext = so.emit_ext; pend_sh = so.emit_sh;
        }
    }
    if (alive) atomicAdd(&ctl->dropped, 1u);     // the bounce guard ended a live path (cannot happen: MAXDEPTH is 100): reported

    if (COUNT) {
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_RAYS, cn.rays);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_BOUNCES, cn.bounces);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_SHADOW, cn.shadow);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_HITS, cn.hits);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_PATHS, cn.paths);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_PRIMS, cn.prims + c_prims);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_WALKED, cn.walk);
        wave_add(ctl->counters[blockIdx.x % kWfShards] + CRT_CNT_NODES, c_nodes);
    }
}

// ------------------------------------------------------------------ resolve
// (one-wave blocks: it runs on the control stream next to the pipes' kernels, and the next batch's queue reset
// is queued behind it -- a block of several waves would wait for as many free slots on one CU)
__global__ __launch_bounds__(64) void k_wf_resolve(const WfParams P, uint32_t last_sample)
{
    const size_t npix = (size_t)P.tw * P.th;
    const size_t pix = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (pix >= npix) return;
    const float4 a4 = P.accum[pix];
    f3 acc = f3{a4.x, a4.y, a4.z};
    const float4 *__restrict__ staging = P.staging[P.batch_id];
    for (uint32_t s = 0; s < P.n_samples; s++) {
        const float4 v = ldnt(&staging[(size_t)s * npix + pix]);
        acc = acc + f3{v.x, v.y, v.z};                           // :108, in sample order
        if (P.frames) {
            // every sample's frame, as the reference shows it (one dispatch + blit per sample, src/main.js:597-620): kept
            // in a ring so that a display loop can show each frame index once while the batches run ahead of it
            const uint32_t smp = last_sample - P.n_samples + 1u + s;
            P.frames[(size_t)((smp - 1u) % P.frame_ring) * npix + pix] = tonemap_rgba8(acc, (float)smp);
        }
    }
    P.accum[pix] = float4{acc.x, acc.y, acc.z, a4.w};
    if (P.n_samples > 0) P.rgba[pix] = tonemap_rgba8(acc, (float)last_sample);
}

// The pixel's RNG seed word tea(px, py*100) (:98) depends on the pixel only: 16 rounds computed once per run
// instead of at every re-arm (where every wave paid for them with a third of its lanes on).
__global__ __launch_bounds__(256) void k_wf_tea(const WfParams P, uint32_t *out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= P.tw * P.th) return;
    const uint32_t lx = i % P.tw, ly = i / P.tw;
    const uint32_t px = P.x0 + lx;
    const uint32_t py = P.y0 + (ly / P.band) * P.band * P.stride + P.phase * P.band + ly % P.band;
    out[i] = tea(px, py * 100u);
}

__global__ void k_wf_init(const WfParams P)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < P.P && !P.keep_pool) P.misc[P.slot_base + i] = uint4{0, 0, 0, 0};
    if (i < kWfShards) {
        WfCtl *c = P.ctl;
        if (!P.keep_pool)
            for (int r = 0; r < 4; r++) {
                for (int k = 0; k < 4; k++) c->shard[r][i].n[k] = 0;
                c->shard[r][i].cur = 0; c->shard[r][i].n_dead = 0;
                for (uint32_t b = 0; b < kWfRing; b++) c->shard[r][i].alive[b] = 0;
            }
        if (i == 0) { if (P.keep_pool) c->side_count[P.batch_id] = 0; else for (uint32_t b = 0; b < kWfRing; b++) c->side_count[b] = 0; }   // (dropped: kept until the host has reported it)
        if (P.reset_wq) {
            P.wq[P.batch_id].work[i].cur = 0;
            if (i == 0) {
                P.wq[P.batch_id].work_done = 0;
                P.wq[P.batch_id].work_per_shard = P.seg[P.batch_id].work_per_shard;
                P.wq[P.batch_id].work_total = P.seg[P.batch_id].work_total;
            }
        }
    }
}

// ------------------------------------------------------------------ launchers
hipError_t wf_launch_init(const WfParams &P, hipStream_t s)
{
    hipLaunchKernelGGL(k_wf_init, dim3(P.keep_pool ? 1u : (P.P + 255) / 256), dim3(256), 0, s, P);
    return hipGetLastError();
}

hipError_t wf_launch_tea(const WfParams &P, uint32_t *out, hipStream_t s)
{
    const size_t npix = (size_t)P.tw * P.th;
    if (npix == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wf_tea, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, P, out);
    return hipGetLastError();
}

hipError_t wf_launch_shade(const WfParams &P, uint32_t it, hipStream_t s)
{
    const uint32_t bps = (4u * P.tail_bound + CRT_WF_SHADE_BLOCK - 1u) / CRT_WF_SHADE_BLOCK;
    const dim3 gs(P.tail_bound ? kWfShards * bps : (P.P + CRT_WF_SHADE_BLOCK - 1) / CRT_WF_SHADE_BLOCK), bs(CRT_WF_SHADE_BLOCK);
    if (P.count) hipLaunchKernelGGL((k_wf_shade<true>), gs, bs, 0, s, P, it);
    else hipLaunchKernelGGL((k_wf_shade<false>), gs, bs, 0, s, P, it);
    return hipGetLastError();
}

hipError_t wf_launch_gen(const WfParams &P, uint32_t it, hipStream_t s)
{
    if (P.gen_blocks == 0u) return hipErrorInvalidValue;
    const dim3 gs(kWfShards * P.gen_blocks), bs(64);
    if (P.count) hipLaunchKernelGGL((k_wf_gen<true>), gs, bs, 0, s, P, it);
    else hipLaunchKernelGGL((k_wf_gen<false>), gs, bs, 0, s, P, it);
    return hipGetLastError();
}

hipError_t wf_launch_trace(const WfParams &P, uint32_t it, uint32_t trace_blocks, hipStream_t s)
{
    const dim3 g(trace_blocks), b(64);
    const int q = !CRT_WF_BVH4 ? 0 : P.sc.nodes8q != nullptr ? 2 : P.sc.nodes4q != nullptr ? 1 : 0;
    if (q == 1 && P.trace_form == 2u) {                          // the regrouped form (quantised 4-wide tree only)
        if (P.count) hipLaunchKernelGGL((k_wf_trace2<true>), g, b, 0, s, P, it);
        else hipLaunchKernelGGL((k_wf_trace2<false>), g, b, 0, s, P, it);
        return hipGetLastError();
    }
    if (P.count) {
        if (q == 2) hipLaunchKernelGGL((k_wf_trace<true, 2>), g, b, 0, s, P, it);
        else if (q == 1) hipLaunchKernelGGL((k_wf_trace<true, 1>), g, b, 0, s, P, it);
        else hipLaunchKernelGGL((k_wf_trace<true, 0>), g, b, 0, s, P, it);
    } else {
        if (q == 2) hipLaunchKernelGGL((k_wf_trace<false, 2>), g, b, 0, s, P, it);
        else if (q == 1) hipLaunchKernelGGL((k_wf_trace<false, 1>), g, b, 0, s, P, it);
        else hipLaunchKernelGGL((k_wf_trace<false, 0>), g, b, 0, s, P, it);
    }
    return hipGetLastError();
}

hipError_t wf_launch_finish(const WfParams &P, WfFinishSegs G, uint32_t max_paths, hipStream_t s)
{
    if (P.tail_bound == 0u || P.tail_bound > 64u || G.n > kWfFinishSegs) return hipErrorInvalidValue;     // paths per wave
    G.blocks_per_seg = (std::min(max_paths, kWfSideCap) + P.tail_bound - 1u) / P.tail_bound;
    if (G.blocks_per_seg == 0 || G.n == 0) return hipSuccess;
    if (P.count) hipLaunchKernelGGL((k_wf_finish<true>), dim3(G.n * G.blocks_per_seg), dim3(64), 0, s, P, G);
    else hipLaunchKernelGGL((k_wf_finish<false>), dim3(G.n * G.blocks_per_seg), dim3(64), 0, s, P, G);
    return hipGetLastError();
}



hipError_t wf_launch_resolve(const WfParams &P, uint32_t last_sample, hipStream_t s)
{
    const size_t npix = (size_t)P.tw * P.th;
    if (npix == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wf_resolve, dim3((unsigned)((npix + 63) / 64)), dim3(64), 0, s, P, last_sample);
    return hipGetLastError();
}

}  // namespace crt
