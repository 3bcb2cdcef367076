// crt_math.h -- the numeric contract of the path tracer, host + device.
//
// The reference shader (ComputeShader.wgsl) leaves FMA contraction, dot/cross
// association and its transcendental builtins implementation-defined.  This
// header fixes ONE conformant choice so that a pixel is a pure function of
// (scene, x, y, sample) on every device and on the host:
//   * +,-,*,/ and sqrt are single IEEE-754 binary32 operations in shader source
//     order  (build with -ffp-contract=off; hipcc's default correctly-rounded
//     f32 divide/sqrt and preserved denormals are relied on);
//   * dot / cross / ray_at are the explicit fma chains below (v_fma_f32);
//   * normalize(v) = v / sqrt(dot(v,v));
//   * sin, cos, exp, log2, exp2 are fixed polynomial kernels (Cephes
//     single-precision coefficients, ~1-2 ulp), pow(x,y) = exp2(y*log2(x)) as
//     the WGSL spec defines it;
//   * max(a,b) = a<b ? b : a, min(a,b) = b<a ? b : a (NaN behaviour fixed).
// DESIGN.md "Numeric contract" is the prose version.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define CRT_HD __host__ __device__ __forceinline__

namespace crt {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

CRT_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
CRT_HD float max_(float a, float b) { return (a < b) ? b : a; }
CRT_HD float min_(float a, float b) { return (b < a) ? b : a; }
CRT_HD float abs_(float a) { return __builtin_fabsf(a); }
CRT_HD float sqrt_(float a) { return __builtin_sqrtf(a); }
CRT_HD float floor_(float a) { return __builtin_floorf(a); }

CRT_HD float bits_f(uint32_t u) { return __builtin_bit_cast(float, u); }
CRT_HD uint32_t f_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

CRT_HD f3 F3(float x, float y, float z) { return f3{x, y, z}; }
CRT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
CRT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
CRT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
CRT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
#ifdef CRT_PROBE_RCP   /* probe build (wrong image): what the vector / scalar divisions cost */
CRT_HD f3 operator/(f3 a, float s) { const float r = 1.0f / s; return f3{a.x * r, a.y * r, a.z * r}; }
#else
CRT_HD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
#endif
CRT_HD float dot(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
CRT_HD f3 cross(f3 a, f3 b) {
    return f3{fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
CRT_HD float length(f3 a) { return sqrt_(dot(a, a)); }
CRT_HD f3 normalize(f3 a) { return a / length(a); }
CRT_HD f3 ray_at(f3 o, f3 d, float t) { return f3{fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z)}; }

CRT_HD f4 F4(float x, float y, float z, float w) { return f4{x, y, z, w}; }
CRT_HD f4 operator+(f4 a, f4 b) { return f4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
CRT_HD f4 operator*(f4 a, f4 b) { return f4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
CRT_HD f4 operator*(f4 a, float s) { return f4{a.x * s, a.y * s, a.z * s, a.w * s}; }
#ifdef CRT_PROBE_RCP
CRT_HD f4 operator/(f4 a, float s) { const float r = 1.0f / s; return f4{a.x * r, a.y * r, a.z * r, a.w * r}; }
#else
CRT_HD f4 operator/(f4 a, float s) { return f4{a.x / s, a.y / s, a.z / s, a.w / s}; }
#endif
CRT_HD float dot(f4 a, f4 b) { return fma_(a.w, b.w, fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x))); }

CRT_HD float pow2i(int n) { return bits_f((uint32_t)(n + 127) << 23); }  // n in [-126,127]

// x >= 0.  Quadrant reduction (3-term Cody-Waite), kernels on [-pi/4, pi/4].
CRT_HD void sincos_(float x, float &s, float &c)
{
    float kf = floor_(fma_(x, 0.63661977236758134f, 0.5f));
    int k = (int)kf;
    float r = fma_(kf, -1.5703125f, x);
    r = fma_(kf, -4.837512969970703125e-4f, r);
    r = fma_(kf, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = fma_(fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float cp = fma_(fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                    z * z, fma_(-0.5f, z, 1.0f));
    int q = k & 3;
    float sv = (q & 1) ? cp : sp;
    float cv = (q & 1) ? sp : cp;
    s = (q & 2) ? -sv : sv;
    c = (q == 1 || q == 2) ? -cv : cv;
}
CRT_HD float sin_(float x) { float s, c; sincos_(x, s, c); return s; }
CRT_HD float cos_(float x) { float s, c; sincos_(x, s, c); return c; }
CRT_HD float tan_(float x) { float s, c; sincos_(x, s, c); return s / c; }

CRT_HD float exp_(float x)
{
    if (x != x) return x;
    if (x > 88.7228394f) return bits_f(0x7F800000u);
    if (x < -103.972084f) return 0.0f;
    float kf = floor_(fma_(x, 1.44269504088896341f, 0.5f));
    float r = fma_(kf, -0.693359375f, x);
    r = fma_(kf, 2.12194440e-4f, r);
    float z = r * r;
    float p = fma_(fma_(fma_(fma_(fma_(1.9875691500e-4f, r, 1.3981999507e-3f), r, 8.3334519073e-3f), r,
                             4.1665795894e-2f), r, 1.6666665459e-1f), r, 5.0000001201e-1f);
    float y = fma_(p, z, r) + 1.0f;
    int k = (int)kf;
    int k1 = (k - (k & 1)) / 2, k2 = k - k1;
    return (y * pow2i(k1)) * pow2i(k2);
}

CRT_HD float log2_(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return bits_f(0x7FC00000u);
    if (x == 0.0f) return bits_f(0xFF800000u);
    if (f_bits(x) == 0x7F800000u) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 16777216.0f; e = -24; }
    uint32_t u = f_bits(x);
    e += (int)(u >> 23) - 126;
    float m = bits_f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = fma_(fma_(fma_(fma_(fma_(fma_(fma_(fma_(7.0376836292e-2f, m, -1.1514610310e-1f), m,
                 1.1676998740e-1f), m, -1.2420140846e-1f), m, 1.4249322787e-1f), m, -1.6668057665e-1f), m,
                 2.0000714765e-1f), m, -2.4999993993e-1f), m, 3.3333331174e-1f);
    y = y * m * z;
    y = fma_(-0.5f, z, y);
    float r = y * 0.44269504088896340735992f;
    r = fma_(m, 0.44269504088896340735992f, r);
    r = r + y;
    r = r + m;
    return r + (float)e;
}

CRT_HD float exp2_(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return bits_f(0x7F800000u);
    if (x < -150.0f) return 0.0f;
    float i0 = floor_(x);
    float r = x - i0;
    if (r > 0.5f) { i0 = i0 + 1.0f; r = r - 1.0f; }
    float p = fma_(fma_(fma_(fma_(fma_(1.535336188319500e-4f, r, 1.339887440266574e-3f), r,
                             9.618437357674640e-3f), r, 5.550332471162809e-2f), r, 2.402264791363012e-1f), r,
                   6.931472028550421e-1f);
    float y = fma_(p, r, 1.0f);
    int k = (int)i0;
    int k1 = (k - (k & 1)) / 2, k2 = k - k1;
    return (y * pow2i(k1)) * pow2i(k2);
}

CRT_HD float pow_(float x, float y)
{
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : bits_f(0x7F800000u);
    return exp2_(y * log2_(x));
}

}  // namespace crt
