// crt_device.h -- device-side scene layout and kernel parameters.
//
// HBM layout (all arrays 16-byte aligned, read-only during a trace):
//   prim   : 3 x float4 per primitive, in BVH leaf order ("slot" order), 48 B:
//              A = (data1.xyz, meta)        meta = cat | mat<<2 | emis<<4 | refl<<18
//              B = (data2.xyz, index bits)  sphere: (r, r*r, 0, index)
//              C = (data3.xyz, e2.e2)       (w used by patches only)
//   primD  : 1 x float4 per primitive: (unit normal of a patch, e1.e1); read for
//            category-0 records only (triangles and spheres never touch it)
//   nodes  : 4 x float4 per inner node, 64 B:
//              (c0.lo.xyz, c0.hi.x) (c0.hi.yz, c1.lo.xy) (c1.lo.z, c1.hi.xyz) (ref0, ref1, -, -)
//   slot_of_index : u32 per primitive (original position -> slot), used by the
//            reference-order loop (CRT_ACCEL_NONE and the NaN-ray fallback)
//   spectra, cie : the reference's tables, unchanged (f32[n][301], f32[3][471])
//   lights : 3 x float4 per light: (data1, emission bits) (data2, index bits) (data3, 1/area)
//   accum  : float4 per tile pixel (xyz + pad; the reference's 16-byte stride)
//   rgba8  : uchar4 per tile pixel
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crt {

constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr int kStackDepth = 32;
constexpr uint32_t kNLambda = 301;
constexpr uint32_t kNCie = 471;

struct DevScene {
    const float4 *prim;
    const float4 *primD;
    const uint32_t *slot_of_index;
    const float4 *nodes;
    const float *spectra;
    const float *cie;
    const float4 *lights;
    uint32_t nprim;
    int32_t root;
    uint32_t nspectra;
    uint32_t nlight;
    float hit_pad;
    float inv_nlight;      // 1.0f / f32(nlight)
    uint32_t W, H;         // full image
    float cam[12];         // llc, horizontal, vertical, eye  (ComputeShader.wgsl:470-487 hoisted)
};

struct TraceParams {
    DevScene sc;
    uint32_t x0, y0, tw, th;        // tile rectangle
    uint32_t first_sample, n_samples;
    float4 *accum;
    uchar4 *rgba;
    unsigned long long *counters;   // CRT_NCOUNTERS, may be null
    uint32_t tiles_x, tiles_y;
};

}  // namespace crt
