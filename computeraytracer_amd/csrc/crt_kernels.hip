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
#include "crt_shade.h"

namespace crt {

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
    const uint32_t px = P.x0 + lx;                             // GLOBAL pixel (seeds, film position)
    const uint32_t py = P.y0 + (ly / P.band) * P.band * P.stride + P.phase * P.band + ly % P.band;
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

            f3 xyzc = spectral_to_xyz(S, radiance, wl);
            acc = acc + xyzc;                                                        // :108
        }

        P.accum[pix] = float4{acc.x, acc.y, acc.z, acc4.w};
        if (P.n_samples > 0) {
            P.rgba[pix] = tonemap_rgba8(acc, (float)(sample - 1u));
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

// ------------------------------------------------------------------ frame assembly (crt_comm.cpp)
// The gathered strips [world][rows_max][W] back into image rows [H][W]: row y of the frame belongs to part
// (y / band) % world and is its local row (y / band / world) * band + y % band (band = 0: contiguous strips of
// ceil(H / world) rows) -- the layout of crt_set_row_bands / crt_layout_rows.  One element = 16 B (accumulator) or 4 B (rgba8).
template <typename T>
__global__ __launch_bounds__(256) void k_assemble(const T *__restrict__ full, T *__restrict__ frame, uint32_t W, uint32_t H, uint32_t world,
                                                  uint32_t rows_max, uint32_t band)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const uint32_t y = (uint32_t)(i / W), x = (uint32_t)(i % W);
    uint32_t part, local;
    if (band) { const uint32_t b = y / band; part = b % world; local = (b / world) * band + y % band; }
    else { const uint32_t per = (H + world - 1u) / world; part = y / per; local = y % per; }
    frame[i] = full[((size_t)part * rows_max + local) * W + x];
}

hipError_t launch_assemble(const void *full, void *frame, uint32_t elem_bytes, uint32_t W, uint32_t H, uint32_t world, uint32_t rows_max,
                           uint32_t band, hipStream_t stream)
{
    const size_t n = (size_t)W * H;
    if (n == 0) return hipSuccess;
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (elem_bytes == 16u) hipLaunchKernelGGL((k_assemble<float4>), g, b, 0, stream, (const float4 *)full, (float4 *)frame, W, H, world, rows_max, band);
    else if (elem_bytes == 4u) hipLaunchKernelGGL((k_assemble<uint32_t>), g, b, 0, stream, (const uint32_t *)full, (uint32_t *)frame, W, H, world, rows_max, band);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace crt
