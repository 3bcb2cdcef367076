/*
 * crt_oracle.c -- TEST INFRASTRUCTURE ONLY (see crt_oracle.h).
 *
 * Plain-C restatement of /root/reference/src/shaders/ComputeShader.wgsl.
 * Function-by-function citations are "CS:<line>" = ComputeShader.wgsl:<line>.
 * The reference loops over every primitive for every ray (CS:503-518); so does
 * this file -- there is no acceleration structure here on purpose.
 *
 * FLOAT SPEC.  WGSL leaves FMA contraction, dot/cross evaluation order and the
 * transcendental builtins implementation-defined.  This restatement fixes one
 * conformant choice (documented in DESIGN.md "Numeric contract"):
 *   - every +,-,*,/ and sqrt is a single IEEE-754 binary32 operation in WGSL
 *     source order (compile with -ffp-contract=off, no fast-math);
 *   - dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))   (dot4 adds one more)
 *   - cross(a,b) = ( fma(a.y,b.z,-(a.z*b.y)), fma(a.z,b.x,-(a.x*b.z)),
 *                    fma(a.x,b.y,-(a.y*b.x)) )
 *   - ray_at(t)  = fma(t, d, o) per component             (CS:304-307)
 *   - normalize(v) = v / sqrt(dot(v,v))  (three IEEE divisions)
 *   - sin, cos, exp, log2, exp2, pow(x,y)=exp2(y*log2 x): the fixed
 *     polynomial kernels below (Cephes single-precision coefficients),
 *     ~1-2 ulp, far inside WGSL's accuracy envelope;
 *   - max(a,b) = a<b ? b : a ;  min(a,b) = b<a ? b : a.
 *
 * Category 2 (triangle) does not exist in the reference; its
 * definition here is this project's own and is the oracle for it.
 */
#include "crt_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ consts */
/* CS:11-20 */
#define PI_F 3.14159265359f
/* CS:12: `const INFINITY : f32 = 0x7F800000` is the INTEGER 2139095040
 * converted to f32 (exactly representable), not +inf. */
#define INFINITY_F 2139095040.0f
#define MAX_U32 0xFFFFFFFFu
#define MAXDEPTH 100u
#define GRID_SIZE 16u
#define LAMBDA_MIN 400.0f
#define LAMBDA_MAX 700.0f
#define MAT_DIFFUSE 0u
#define MAT_LIGHT 1u
#define MAT_GLASS 2u
#define NLAMBDA 301u
#define NCIE 471u

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline float fma_(float a, float b, float c) { return fmaf(a, b, c); }
/* Sensitivity variants (oracle/Makefile `variants`, oracle/sensitivity.py): other CONFORMANT WGSL evaluations of the
 * same shader, to measure how far a real WebGPU stack may sit from the pinned contract.  Never used by the tests'
 * parity checks.  ORC_VARIANT_UNFUSED: dot / cross / ray_at as separate multiplies and adds, left to right.
 * ORC_VARIANT_LIBM: sin cos tan exp pow from libm instead of the fixed kernels. */
#ifdef ORC_VARIANT_UNFUSED
static inline float gfma_(float a, float b, float c) { float p = a * b; return p + c; }
#else
static inline float gfma_(float a, float b, float c) { return fmaf(a, b, c); }
#endif
static inline float max_(float a, float b) { return (a < b) ? b : a; }
static inline float min_(float a, float b) { return (b < a) ? b : a; }
static inline float abs_(float a) { return fabsf(a); }

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3s(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 div3s(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return gfma_(a.z, b.z, gfma_(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(gfma_(a.y, b.z, -(a.z * b.y)), gfma_(a.z, b.x, -(a.x * b.z)),
              gfma_(a.x, b.y, -(a.y * b.x)));
}
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) { return div3s(a, length3(a)); }

static inline v4 V4(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v4 mul4(v4 a, v4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 mul4s(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline v4 div4s(v4 a, float s) { return V4(a.x / s, a.y / s, a.z / s, a.w / s); }
static inline v4 add4(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline float dot4(v4 a, v4 b) {
    return gfma_(a.w, b.w, gfma_(a.z, b.z, gfma_(a.y, b.y, a.x * b.x)));
}

/* ------------------------------------------------------- deterministic math */
static inline float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float pow2i(int n) { return bits_f((uint32_t)(n + 127) << 23); } /* n in [-126,127] */

/* sin & cos, x >= 0 (the live path only evaluates [0, 2*pi], CS:755-757, and
 * fov/2, CS:479).  3-term Cody-Waite reduction to [-pi/4,pi/4], Cephes kernels. */
static __attribute__((unused)) void sincos_(float x, float *s, float *c)
{
    float kf = floorf(fma_(x, 0.63661977236758134f, 0.5f));
    int k = (int)kf;
    float r = fma_(kf, -1.5703125f, x);
    r = fma_(kf, -4.837512969970703125e-4f, r);
    r = fma_(kf, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = fma_(fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f),
                    z * r, r);
    float cp = fma_(fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                         4.166664568298827e-2f),
                    z * z, fma_(-0.5f, z, 1.0f));
    switch (k & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}
#ifdef ORC_VARIANT_LIBM
static void sincos_libm_(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
#define sincos_ sincos_libm_
static float sin_(float x) { return sinf(x); }
static float cos_(float x) { return cosf(x); }
static float tan_(float x) { return tanf(x); }
#else
static float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
static float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }
static float tan_(float x) { float s, c; sincos_(x, &s, &c); return s / c; }
#endif

static float exp_(float x)
{
#ifdef ORC_VARIANT_LIBM
    return expf(x);
#endif
    if (x != x) return x;
    if (x > 88.7228394f) return bits_f(0x7F800000u);
    if (x < -103.972084f) return 0.0f;
    float kf = floorf(fma_(x, 1.44269504088896341f, 0.5f));
    float r = fma_(kf, -0.693359375f, x);
    r = fma_(kf, 2.12194440e-4f, r);
    float z = r * r;
    float p = fma_(fma_(fma_(fma_(fma_(1.9875691500e-4f, r, 1.3981999507e-3f), r,
                                  8.3334519073e-3f), r, 4.1665795894e-2f), r,
                        1.6666665459e-1f), r, 5.0000001201e-1f);
    float y = fma_(p, z, r) + 1.0f;
    int k = (int)kf;
    int k1 = (k - (k & 1)) / 2, k2 = k - k1;
    return (y * pow2i(k1)) * pow2i(k2);
}

static float log2_(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return bits_f(0x7FC00000u);
    if (x == 0.0f) return bits_f(0xFF800000u);
    if (f_bits(x) == 0x7F800000u) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 16777216.0f; e = -24; }
    uint32_t u = f_bits(x);
    e += (int)(u >> 23) - 126;                     /* x = m * 2^e, m in [0.5,1) */
    float m = bits_f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = fma_(fma_(fma_(fma_(fma_(fma_(fma_(fma_(7.0376836292e-2f, m, -1.1514610310e-1f),
                 m, 1.1676998740e-1f), m, -1.2420140846e-1f), m, 1.4249322787e-1f), m,
                 -1.6668057665e-1f), m, 2.0000714765e-1f), m, -2.4999993993e-1f), m,
                 3.3333331174e-1f);
    y = y * m * z;
    y = fma_(-0.5f, z, y);
    /* log2(1+m) = (m+y)*log2(e), split as Cephes log2f does */
    float r = y * 0.44269504088896340735992f;
    r = fma_(m, 0.44269504088896340735992f, r);
    r = r + y;
    r = r + m;
    return r + (float)e;
}

static float exp2_(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return bits_f(0x7F800000u);
    if (x < -150.0f) return 0.0f;
    float i0 = floorf(x);
    float r = x - i0;
    if (r > 0.5f) { i0 = i0 + 1.0f; r = r - 1.0f; }
    float p = fma_(fma_(fma_(fma_(fma_(1.535336188319500e-4f, r, 1.339887440266574e-3f), r,
                             9.618437357674640e-3f), r, 5.550332471162809e-2f), r,
                        2.402264791363012e-1f), r, 6.931472028550421e-1f);
    float y = fma_(p, r, 1.0f);
    int k = (int)i0;
    int k1 = (k - (k & 1)) / 2, k2 = k - k1;
    return (y * pow2i(k1)) * pow2i(k2);
}

/* WGSL pow(x,y) is specified as exp2(y*log2(x)); that is what this is. */
static float pow_(float x, float y)
{
#ifdef ORC_VARIANT_LIBM
    return powf(x, y);
#endif
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : bits_f(0x7F800000u);
    return exp2_(y * log2_(x));
}

void orc_math_eval(int fn, const float *a, const float *b, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        switch (fn) {
        case ORC_FN_SIN: out[i] = sin_(a[i]); break;
        case ORC_FN_COS: out[i] = cos_(a[i]); break;
        case ORC_FN_EXP: out[i] = exp_(a[i]); break;
        case ORC_FN_LOG2: out[i] = log2_(a[i]); break;
        case ORC_FN_EXP2: out[i] = exp2_(a[i]); break;
        case ORC_FN_POW: out[i] = pow_(a[i], b[i]); break;
        case ORC_FN_SQRT: out[i] = sqrtf(a[i]); break;
        case ORC_FN_DIV: out[i] = a[i] / b[i]; break;
        case ORC_FN_TAN: out[i] = tan_(a[i]); break;
        default: out[i] = 0.0f;
        }
    }
}

/* ---------------------------------------------------------------------- RNG */
/* CS:865-877 */
uint32_t orc_tea(uint32_t val0, uint32_t val1)
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (int n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

typedef struct { uint32_t x, y, z, w; } u4;

/* CS:879-891 */
static inline void pcg4d(u4 *s)
{
    s->x = s->x * 1664525u + 1013904223u;
    s->y = s->y * 1664525u + 1013904223u;
    s->z = s->z * 1664525u + 1013904223u;
    s->w = s->w * 1664525u + 1013904223u;
    s->x += s->y * s->w;
    s->y += s->z * s->x;
    s->z += s->x * s->y;
    s->w += s->y * s->z;
    s->x ^= s->x >> 16; s->y ^= s->y >> 16; s->z ^= s->z >> 16; s->w ^= s->w >> 16;
    s->x += s->y * s->w;
    s->y += s->z * s->x;
    s->z += s->x * s->y;
    s->w += s->y * s->z;
}

void orc_rand_kat(uint32_t x, uint32_t y, uint32_t sample, uint32_t n, uint32_t *out24,
                  uint32_t seed_out[4])
{
    u4 s = { y, x * 100u, sample, orc_tea(x, y * 100u) }; /* CS:98 */
    for (uint32_t i = 0; i < n; i++) { pcg4d(&s); out24[i] = s.x & 0x00ffffffu; }
    seed_out[0] = s.x; seed_out[1] = s.y; seed_out[2] = s.z; seed_out[3] = s.w;
}

/* --------------------------------------------------------------- scene view */
typedef struct {
    uint32_t category;
    v3 data1, data2, data3;
    uint32_t emission, reflectance, material, index;
} prim_t;

/* 80-byte record: category@0, data1@16, data2@32, data3@48, data4@64 (CS:41-47,
 * main.js:211-246).  Copied by value like the WGSL does. */
static inline prim_t load_prim(const uint8_t *base, uint32_t i)
{
    prim_t p;
    const uint8_t *r = base + (size_t)i * 80u;
    float f[12]; uint32_t u[4];
    memcpy(&p.category, r, 4);
    memcpy(f, r + 16, 12); memcpy(f + 3, r + 32, 12); memcpy(f + 6, r + 48, 12);
    memcpy(u, r + 64, 16);
    p.data1 = V3(f[0], f[1], f[2]); p.data2 = V3(f[3], f[4], f[5]); p.data3 = V3(f[6], f[7], f[8]);
    p.emission = u[0]; p.reflectance = u[1]; p.material = u[2]; p.index = u[3];
    return p;
}

typedef struct { v3 origin, direction; } ray_t;
typedef struct { v3 position, normal; uint32_t emission_index, reflectance_index, material; } shape_isect;
typedef struct { float t_min, t_max; uint32_t index, exclude; int hit; v3 ray_origin, ray_direction; } isect_ctx;
typedef struct { shape_isect si; isect_ctx ctx; } isect_t;

typedef struct {
    const orc_scene *sc;
    u4 seed;
    uint32_t sample;
    float hit_pad;
    uint64_t c_rays, c_tests, c_bounces, c_shadow, c_rand;
    float *raylog;            /* optional: 12 floats per intersect() call */
    uint32_t raylog_cap, raylog_n;
} tstate;

/* CS:893-897 */
static inline float rand_(tstate *ts)
{
    pcg4d(&ts->seed);
    ts->c_rand++;
    return (float)(ts->seed.x & 0x00ffffffu) / 16777216.0f;
}

/* --------------------------------------------------------- scene-scale pad */
/* Category-2 acceptance rule: a triangle hit is kept only if the hit point
 * lies inside the triangle's AABB grown by hit_pad = 2^-17 * S, S = largest
 * |coordinate| over every primitive corner and the camera eye.  (Makes the
 * brute-force answer provably reachable by a BVH with padded boxes.) */
float orc_hit_pad(const orc_scene *sc)
{
    float S = 0.0f;
    for (uint32_t i = 0; i < sc->nprim; i++) {
        prim_t p = load_prim(sc->primitives, i);
        v3 c[4]; int nc = 0;
        if (p.category == 1u) {
            float r = abs_(p.data2.x);
            c[0] = V3(p.data1.x - r, p.data1.y - r, p.data1.z - r);
            c[1] = V3(p.data1.x + r, p.data1.y + r, p.data1.z + r);
            nc = 2;
        } else {
            c[0] = p.data1; c[1] = add3(p.data1, p.data2); c[2] = add3(p.data1, p.data3);
            nc = 3;
            if (p.category == 0u) { c[3] = add3(c[1], p.data3); nc = 4; }
        }
        for (int k = 0; k < nc; k++) {
            S = max_(S, abs_(c[k].x)); S = max_(S, abs_(c[k].y)); S = max_(S, abs_(c[k].z));
        }
    }
    S = max_(S, abs_(sc->camera[0])); S = max_(S, abs_(sc->camera[1])); S = max_(S, abs_(sc->camera[2]));
    return S * 7.62939453125e-06f; /* 2^-17 */
}

/* ------------------------------------------------------------ intersection */
/* CS:709-721 */
static inline isect_ctx create_ctx(uint32_t exclude)
{
    isect_ctx c;
    c.t_min = 0.001f; c.t_max = INFINITY_F; c.index = MAX_U32; c.exclude = exclude; c.hit = 0;
    c.ray_origin = V3(0, 0, 0); c.ray_direction = V3(0, 0, 0);
    return c;
}

/* CS:520-632 (+ category 2, this project's own) */
static void ray_intersection(const prim_t *pr, const ray_t *ray, isect_ctx *ctx, shape_isect *si,
                             float hit_pad)
{
    uint32_t category = pr->category;
    if (category == 0u) {                                   /* CS:525-583 */
        if (ctx->exclude == pr->index) return;               /* CS:527-532 */
        v3 edge1 = pr->data2, edge2 = pr->data3;
        v3 normal = normalize3(cross3(edge1, edge2));        /* CS:536 */
        v3 direction = ray->direction;
        float ndotd = dot3(normal, direction);
        if (ndotd > 0.0f) normal = neg3(normal);             /* CS:541-544 */
        ndotd = dot3(normal, direction);
        if (abs_(ndotd) < 0.0001f) return;                   /* CS:546 */
        v3 origin = ray->origin;
        v3 oo = sub3(pr->data1, origin);
        float t = dot3(normal, oo) / ndotd;                  /* CS:554 */
        if (t < ctx->t_min || t > ctx->t_max) return;        /* CS:557 (tie: later wins, Q4) */
        v3 p = V3(gfma_(t, direction.x, origin.x), gfma_(t, direction.y, origin.y),
                  gfma_(t, direction.z, origin.z));          /* CS:561 ray_at */
        v3 m = sub3(p, pr->data1);
        float u = dot3(m, edge1) / dot3(edge1, edge1);       /* CS:563 */
        float v = dot3(m, edge2) / dot3(edge2, edge2);       /* CS:564 */
        if (u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f) return;
        si->position = p; si->normal = normal;
        si->emission_index = pr->emission; si->reflectance_index = pr->reflectance;
        si->material = pr->material;
        ctx->t_max = t; ctx->index = pr->index; ctx->hit = 1;
        ctx->ray_origin = origin; ctx->ray_direction = direction;
        return;
    }
    if (category == 1u) {                                   /* CS:584-631 */
        if (ctx->exclude == pr->index) return;
        v3 center = pr->data1;
        float radius = pr->data2.x;
        float radius_squared = radius * radius;
        v3 origin = ray->origin, direction = ray->direction;
        v3 co = sub3(origin, center);
        float a = dot3(direction, direction);
        float b = 2.0f * dot3(direction, co);
        float c = dot3(co, co) - radius_squared;
        float disc = b * b - 4.0f * a * c;                   /* CS:601 */
        if (disc <= 0.0f) return;
        float sq = sqrtf(disc);
        float t = (-b - sq) / (2.0f * a);
        if (t < ctx->t_min || t > ctx->t_max) {
            t = (-b + sq) / (2.0f * a);
            if (t < ctx->t_min || t > ctx->t_max) return;
        }
        v3 p = V3(gfma_(t, direction.x, origin.x), gfma_(t, direction.y, origin.y),
                  gfma_(t, direction.z, origin.z));
        v3 normal = normalize3(sub3(p, center));             /* CS:618 outward always (Q6) */
        si->position = p; si->normal = normal;
        si->emission_index = pr->emission; si->reflectance_index = pr->reflectance;
        si->material = pr->material;
        ctx->t_max = t; ctx->index = pr->index; ctx->hit = 1;
        ctx->ray_origin = origin; ctx->ray_direction = direction;
        return;
    }
    if (category == 2u) {
        /* Triangle v0=data1, e1=data2 (v1-v0), e2=data3 (v2-v0).  Moeller-
         * Trumbore with the reference's conventions: same t window and
         * "equal t, later wins" rule, exclude by index, normal flipped to face
         * the ray like a patch.  NaN-safe (every test is an accept-form). */
        if (ctx->exclude == pr->index) return;
        v3 v0 = pr->data1, e1 = pr->data2, e2 = pr->data3;
        v3 origin = ray->origin, d = ray->direction;
        v3 pvec = cross3(d, e2);
        float det = dot3(e1, pvec);
        if (det == 0.0f) return;
        float inv = 1.0f / det;
        v3 tvec = sub3(origin, v0);
        float u = dot3(tvec, pvec) * inv;
        if (!(u >= 0.0f && u <= 1.0f)) return;
        v3 qvec = cross3(tvec, e1);
        float v = dot3(d, qvec) * inv;
        if (!(v >= 0.0f && (u + v) <= 1.0f)) return;
        float t = dot3(e2, qvec) * inv;
        if (!(t >= ctx->t_min && t <= ctx->t_max)) return;
        v3 p = V3(gfma_(t, d.x, origin.x), gfma_(t, d.y, origin.y), gfma_(t, d.z, origin.z));
        v3 v1 = add3(v0, e1), v2 = add3(v0, e2);
        v3 lo = V3(min_(v0.x, min_(v1.x, v2.x)) - hit_pad, min_(v0.y, min_(v1.y, v2.y)) - hit_pad,
                   min_(v0.z, min_(v1.z, v2.z)) - hit_pad);
        v3 hi = V3(max_(v0.x, max_(v1.x, v2.x)) + hit_pad, max_(v0.y, max_(v1.y, v2.y)) + hit_pad,
                   max_(v0.z, max_(v1.z, v2.z)) + hit_pad);
        if (!(p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z &&
              p.z <= hi.z)) return;
        v3 normal = normalize3(cross3(e1, e2));
        if (dot3(normal, d) > 0.0f) normal = neg3(normal);
        si->position = p; si->normal = normal;
        si->emission_index = pr->emission; si->reflectance_index = pr->reflectance;
        si->material = pr->material;
        ctx->t_max = t; ctx->index = pr->index; ctx->hit = 1;
        ctx->ray_origin = origin; ctx->ray_direction = d;
        return;
    }
}

/* CS:503-518.  `shadow` is always false on the live path (CS:135, CS:699). */
static isect_t intersect(tstate *ts, const ray_t *ray, uint32_t exclude)
{
    isect_t r;
    memset(&r.si, 0, sizeof r.si);                           /* WGSL zero-inits `var` */
    r.ctx = create_ctx(exclude);
    const orc_scene *sc = ts->sc;
    for (uint32_t i = 0; i < sc->nprim; i++) {
        prim_t p = load_prim(sc->primitives, i);
        ray_intersection(&p, ray, &r.ctx, &r.si, ts->hit_pad);
    }
    ts->c_rays++;
    ts->c_tests += sc->nprim;
    if (ts->raylog && ts->raylog_n < ts->raylog_cap) {
        float *l = ts->raylog + (size_t)ts->raylog_n * 12;
        uint32_t idx = r.ctx.hit ? r.ctx.index : MAX_U32;
        l[0] = ray->origin.x; l[1] = ray->origin.y; l[2] = ray->origin.z;
        l[3] = ray->direction.x; l[4] = ray->direction.y; l[5] = ray->direction.z;
        memcpy(&l[6], &exclude, 4); memcpy(&l[7], &idx, 4);
        l[8] = r.ctx.t_max; l[9] = r.si.normal.x; l[10] = r.si.normal.y; l[11] = r.si.normal.z;
        ts->raylog_n++;
    }
    return r;
}

/* CS:697-705 */
static isect_t shadow_intersect(tstate *ts, const ray_t *ray, uint32_t include, uint32_t exclude)
{
    isect_t r = intersect(ts, ray, exclude);
    ts->c_shadow++;
    if (r.ctx.index != include) r.ctx.hit = 0;
    return r;
}

/* ------------------------------------------------------------------ spectra */
/* CS:310-313 */
static inline v4 sample_spectrum(const orc_scene *sc, uint32_t index, const uint32_t l[4])
{
    /* OOB storage reads clamp (WebGPU robust access; same pin as Q7). */
    if (index >= sc->nspectra) index = sc->nspectra - 1u;
    const float *row = sc->spectra + (size_t)index * NLAMBDA;
    return V4(row[l[0]], row[l[1]], row[l[2]], row[l[3]]);
}

/* CS:315-322 */
static inline void sample_wavelengths(tstate *ts, uint32_t l[4])
{
    float u = rand_(ts);
    uint32_t range = (uint32_t)(LAMBDA_MAX - LAMBDA_MIN) + 1u;
    /* mix(0, 301, u) = 0*(1-u) + 301*u = 301*u exactly */
    uint32_t lambda = (uint32_t)((LAMBDA_MAX - LAMBDA_MIN + 1.0f) * u);
    l[0] = lambda; l[1] = (lambda + 4u) % range; l[2] = (lambda + 8u) % range;
    l[3] = (lambda + 12u) % range;
}

/* ------------------------------------------------------------------- lights */
static inline prim_t light_at(const orc_scene *sc, uint32_t idx)
{
    /* Q7: OOB runtime-array read -> clamp to arrayLength-1 (Tint/Dawn). */
    if (idx >= sc->nlight) idx = sc->nlight - 1u;
    return load_prim(sc->lights, idx);
}

/* CS:297-302 */
static inline float power_heuristic(float nf, float f_pdf, float ng, float g_pdf)
{
    float f = nf * f_pdf, g = ng * g_pdf;
    return (f * f) / (f * f + g * g);
}

/* CS:357-377 */
static float compute_light_pdf(const orc_scene *sc, const isect_t *is)
{
    prim_t light = light_at(sc, is->si.emission_index);      /* Q7: spectrum idx as light idx */
    float light_area = length3(light.data2) * length3(light.data3);
    float light_area_pdf = 1.0f / light_area;
    float abs_cos_theta = max_(0.00001f, abs_(dot3(is->si.normal, neg3(is->ctx.ray_direction))));
    float distance = length3(sub3(is->si.position, is->ctx.ray_origin));
    float distance_squared = pow_(distance, 2.0f);           /* CS:368 */
    float geometric_term = abs_cos_theta / distance_squared;
    float light_solid_angle_pdf = light_area_pdf / geometric_term;
    float number_of_lights = (float)sc->nlight;
    float light_selection_pdf = 1.0f / number_of_lights;
    return light_selection_pdf * light_solid_angle_pdf;
}

/* CS:379-408 */
static v4 compute_light_radiance(tstate *ts, const isect_t *is, const uint32_t wl[4])
{
    const orc_scene *sc = ts->sc;
    /* sample_lights CS:341-347 */
    float u0 = rand_(ts);
    float range = (float)sc->nlight;
    uint32_t li = (uint32_t)(range * u0);                    /* mix(0,range,u) */
    prim_t light = light_at(sc, li);
    /* sample_light CS:349-355 */
    float u = rand_(ts);
    float v = rand_(ts);
    v3 pl = add3(add3(light.data1, mul3s(light.data2, u)), mul3s(light.data3, v));
    v3 light_dir = normalize3(sub3(pl, is->si.position));
    ray_t sray = { is->si.position, light_dir };
    isect_t sh = shadow_intersect(ts, &sray, light.index, is->ctx.index);
    float cos_theta = max_(0.0f, dot3(is->si.normal, light_dir));
    v4 spec = sample_spectrum(sc, light.emission, wl);
    v4 le = mul4s(spec, cos_theta);
    float pdf_l = compute_light_pdf(sc, &sh);                /* evaluated before the hit test */
    if (sh.ctx.hit) {
        float pdf_b = cos_theta / PI_F;
        float weight_l = power_heuristic(1.0f, pdf_l, 1.0f, pdf_b);
        return div4s(mul4s(le, weight_l), pdf_l);            /* CS:400 */
    }
    return V4(0, 0, 0, 0);
}

/* ---------------------------------------------------------------- materials */
/* CS:751-774 */
static v3 cosine_weighted_sample_hemisphere(tstate *ts, v3 normal, float *pdf)
{
    float u = rand_(ts);
    float v = rand_(ts);
    float r = sqrtf(u);
    float theta = (2.0f * PI_F) * v;
    float st, ct;
    sincos_(theta, &st, &ct);
    float x = r * ct;
    float y = r * st;
    float z = sqrtf(max_(0.0f, 1.0f - u));
    v3 up = (abs_(normal.z) < 0.999f) ? V3(0, 0, 1) : V3(1, 0, 0);
    v3 tangent = normalize3(cross3(up, normal));
    v3 bitangent = cross3(normal, tangent);
    v3 dir = add3(add3(mul3s(tangent, x), mul3s(bitangent, y)), mul3s(normal, z));
    *pdf = z / PI_F;
    return dir;
}

/* CS:814-837 */
static float fresnel_s(v3 ray_dir, v3 normal, float eta1, float eta2)
{
    float cosi = min_(max_(dot3(ray_dir, normal), -1.0f), 1.0f); /* clamp */
    float eta = eta1 / eta2;
    if (cosi > 0.0f) eta = eta2 / eta1;
    float sint2 = eta * eta * (1.0f - cosi * cosi);
    if (sint2 > 1.0f) return 1.0f;
    float cost = sqrtf(1.0f - sint2);
    cosi = abs_(cosi);
    float Rs = ((eta1 * cosi) - (eta2 * cost)) / ((eta1 * cosi) + (eta2 * cost)); /* Q9: unswapped */
    float Rp = ((eta2 * cosi) - (eta1 * cost)) / ((eta2 * cosi) + (eta1 * cost));
    return (Rs * Rs + Rp * Rp) / 2.0f;
}

/* WGSL builtins reflect / refract */
static inline v3 reflect_(v3 e1, v3 e2)
{
    float k = 2.0f * dot3(e2, e1);
    return sub3(e1, mul3s(e2, k));
}
static inline v3 refract_(v3 e1, v3 e2, float e3)
{
    float d = dot3(e2, e1);
    float k = 1.0f - e3 * e3 * (1.0f - d * d);
    if (k < 0.0f) return V3(0, 0, 0);
    float s = e3 * d + sqrtf(k);
    return sub3(mul3s(e1, e3), mul3s(e2, s));
}

/* ---------------------------------------------------------------- path trace */
/* CS:119-295 */
static v4 path_trace(tstate *ts, ray_t ray, const uint32_t wl[4], orc_transcript *tr)
{
    const orc_scene *sc = ts->sc;
    uint32_t depth = 0;
    v4 accumulated_radiance = V4(0, 0, 0, 0);
    v4 beta = V4(1, 1, 1, 1);
    float last_bounce_pdf = 1.0f;
    uint32_t exclude = MAX_U32;
    v4 BRDF = V4(1, 1, 1, 1);
    int specular_bounce = 0;
    float etaScale = 1.0f;
    int inTransmission = 0;

    for (;;) {
        isect_t is = intersect(ts, &ray, exclude);           /* CS:135 */
        ts->c_bounces++;
        if (tr && tr->n_hits < 128) tr->hits[tr->n_hits++] = is.ctx.hit ? is.ctx.index : MAX_U32;
        if (!is.ctx.hit) break;                              /* CS:141 */
        exclude = is.ctx.index;                              /* CS:146 */
        uint32_t material = is.si.material;
        if (material == MAT_LIGHT) {                         /* CS:149-164 */
            v4 le = sample_spectrum(sc, is.si.emission_index, wl);
            if (depth == 0 || specular_bounce) {
                accumulated_radiance = add4(accumulated_radiance, mul4(beta, le));
            } else {
                float pdf_l = compute_light_pdf(sc, &is);
                float weight_b = power_heuristic(1.0f, last_bounce_pdf, 1.0f, pdf_l);
                accumulated_radiance = add4(accumulated_radiance, mul4(mul4s(le, weight_b), beta));
            }
            break;
        }
        if (depth >= MAXDEPTH) break;                        /* CS:167 */
        if (inTransmission) {                                /* CS:173-179 */
            float distance = length3(sub3(is.ctx.ray_origin, is.si.position));
            v4 ext = sample_spectrum(sc, sc->nspectra - 1u, wl);
            v4 att = V4(exp_(-ext.x * distance), exp_(-ext.y * distance), exp_(-ext.z * distance),
                        exp_(-ext.w * distance));
            beta = mul4(beta, att);
        }
        if (material == MAT_DIFFUSE) {                       /* CS:182-204 */
            BRDF = div4s(sample_spectrum(sc, is.si.reflectance_index, wl), PI_F);
            v4 le = compute_light_radiance(ts, &is, wl);
            accumulated_radiance = add4(accumulated_radiance, mul4(mul4(BRDF, le), beta));
            v3 new_direction = cosine_weighted_sample_hemisphere(ts, is.si.normal, &last_bounce_pdf);
            float cos_theta = abs_(dot3(is.si.normal, new_direction));
            beta = mul4(beta, div4s(mul4s(BRDF, cos_theta), last_bounce_pdf));
            ray.origin = is.si.position;
            ray.direction = new_direction;
            specular_bounce = 0;
        }
        if (material == MAT_GLASS) {                         /* CS:208-276 */
            float eta1 = 1.0f, eta2 = 1.5f;
            float eta = eta1 / eta2;
            float cos_theta = dot3(is.si.normal, ray.direction);
            float reflected = fresnel_s(ray.direction, is.si.normal, eta1, eta2);
            float pr = reflected;
            float pt = 1.0f - reflected;
            float u = rand_(ts);
            v3 current_direction = ray.direction;
            v3 current_normal = is.si.normal;
            if (cos_theta > 0.0f) {
                eta = 1.0f / eta;
                current_normal = neg3(current_normal);
            }
            v3 new_direction;
            if (u < pr / (pr + pt)) {                        /* CS:238 */
                new_direction = reflect_(current_direction, current_normal);
                ray.origin = is.si.position;
                specular_bounce = 1;
                exclude = MAX_U32;
            } else {
                specular_bounce = 1;
                exclude = MAX_U32;
                new_direction = refract_(current_direction, current_normal, eta);
                new_direction = normalize3(new_direction);
                ray.origin = is.si.position;
                beta = mul4s(beta, eta * eta);
                etaScale = etaScale / (eta * eta);
                inTransmission = !inTransmission;
            }
            ray.direction = new_direction;
        }
        /* CS:279-289 (Q11: xyz of the 4-wavelength beta only) */
        v4 rbeta = mul4s(beta, etaScale);
        float max_beta_component = max_(rbeta.x, max_(rbeta.y, rbeta.z));
        if (depth > 1u && max_beta_component < 1.0f) {
            float q = max_(0.0f, 1.0f - max_beta_component);
            if (rand_(ts) < q) break;
            beta = div4s(beta, 1.0f - q);
        }
        depth++;
    }
    return accumulated_radiance;
}

/* ------------------------------------------------------------------- camera */
/* CS:470-487: everything that does not depend on the pixel. */
void orc_camera_frame(const float cam[16], float out[12])
{
    v3 eye = V3(cam[0], cam[1], cam[2]), lookat = V3(cam[4], cam[5], cam[6]), up = V3(cam[8], cam[9], cam[10]);
    float vw_px = cam[11], vh_px = cam[12], focal = cam[13];
    v3 w = normalize3(sub3(eye, lookat));
    v3 u = normalize3(cross3(up, w));
    v3 v = cross3(w, u);
    float aspect_ratio = vw_px / vh_px;
    float viewport_height = 2.0f * tan_(focal / 2.0f);
    float viewport_width = aspect_ratio * viewport_height;
    v3 horizontal = mul3s(u, viewport_width);
    v3 vertical = mul3s(v, viewport_height);
    v3 llc = sub3(sub3(sub3(eye, div3s(horizontal, 2.0f)), div3s(vertical, 2.0f)), w);
    out[0] = llc.x; out[1] = llc.y; out[2] = llc.z;
    out[3] = horizontal.x; out[4] = horizontal.y; out[5] = horizontal.z;
    out[6] = vertical.x; out[7] = vertical.y; out[8] = vertical.z;
    out[9] = eye.x; out[10] = eye.y; out[11] = eye.z;
}

/* CS:477-500 */
static ray_t camera_ray(tstate *ts, uint32_t px, uint32_t py, uint32_t W, uint32_t H, const float fr[12])
{
    float jx = rand_(ts);                                    /* CS:497 */
    float s = ((float)px + ((float)(ts->sample % GRID_SIZE) + jx) / (float)GRID_SIZE) / (float)W;
    float jy = rand_(ts);                                    /* CS:498 */
    float t = ((float)H - (float)py + ((float)(ts->sample % GRID_SIZE) + jy) / (float)GRID_SIZE) / (float)H;
    v3 llc = V3(fr[0], fr[1], fr[2]), hor = V3(fr[3], fr[4], fr[5]), ver = V3(fr[6], fr[7], fr[8]);
    v3 eye = V3(fr[9], fr[10], fr[11]);
    v3 d = sub3(add3(add3(llc, mul3s(hor, s)), mul3s(ver, t)), eye);
    ray_t r = { eye, normalize3(d) };
    return r;
}

/* ------------------------------------------------------------------- colour */
/* CS:419-426 */
static v3 spectral_to_xyz(const orc_scene *sc, v4 radiance, const uint32_t wl[4])
{
    const float *X = sc->cie, *Y = sc->cie + NCIE, *Z = sc->cie + 2 * NCIE;
    v4 xb = V4(X[wl[0] + 40], X[wl[1] + 40], X[wl[2] + 40], X[wl[3] + 40]);
    v4 yb = V4(Y[wl[0] + 40], Y[wl[1] + 40], Y[wl[2] + 40], Y[wl[3] + 40]);
    v4 zb = V4(Z[wl[0] + 40], Z[wl[1] + 40], Z[wl[2] + 40], Z[wl[3] + 40]);
    v3 xyz = V3(dot4(xb, radiance), dot4(yb, radiance), dot4(zb, radiance));
    float integ = 106.856895f;
    float lambda_range = LAMBDA_MAX - LAMBDA_MIN;
    return div3s(mul3s(xyz, lambda_range), integ * 4.0f);
}

/* CS:428-434 */
static v3 xyz_to_linear_rgb(v3 c)
{
    float r = 3.2404542f * c.x + -1.5371385f * c.y + -0.4985314f * c.z;
    float g = -0.9692660f * c.x + 1.8760108f * c.y + 0.0415560f * c.z;
    float b = 0.0556434f * c.x + -0.2040259f * c.y + 1.0572252f * c.z;
    return V3(r, g, b);
}

/* CS:436-439 */
static v3 tone_map(v3 rgb, float exposure)
{
    return V3(1.0f - exp_(-rgb.x * exposure), 1.0f - exp_(-rgb.y * exposure),
              1.0f - exp_(-rgb.z * exposure));
}

/* CS:441-467, including the G-channel bug (Q10) */
static void gamma_correct(v3 *rgb)
{
    if (rgb->x < 0.0031308f) rgb->x *= 12.92f;
    else rgb->x = 1.055f * pow_(rgb->x, (float)(1.0 / 2.4)) - 0.055f;
    if (rgb->y < 0.0031308f) rgb->y *= 12.92f * rgb->y;
    else rgb->y = 1.055f * pow_(rgb->y, (float)(1.0 / 2.4)) - 0.055f;
    if (rgb->z < 0.0031308f) rgb->z = 12.92f * rgb->z;
    else rgb->z = 1.055f * pow_(rgb->z, (float)(1.0 / 2.4)) - 0.055f;
}

/* textureStore to rgba8unorm: clamp to [0,1], scale, round to nearest. NaN -> 0 */
static inline uint8_t unorm8(float x)
{
    if (!(x > 0.0f)) return 0;
    if (x > 1.0f) x = 1.0f;
    return (uint8_t)(x * 255.0f + 0.5f);
}

/* ---------------------------------------------------------------- one pixel */
/* CS:77-117 for one (pixel, sample) */
static v3 pixel_sample(tstate *ts, uint32_t x, uint32_t y, uint32_t W, uint32_t H,
                       const float frame[12], orc_transcript *tr)
{
    ts->seed.x = y; ts->seed.y = x * 100u; ts->seed.z = ts->sample;
    ts->seed.w = orc_tea(x, y * 100u);                       /* CS:98 */
    ray_t ray = camera_ray(ts, x, y, W, H, frame);
    uint32_t wl[4];
    sample_wavelengths(ts, wl);
    v4 radiance = path_trace(ts, ray, wl, tr);
    v3 xyz = spectral_to_xyz(ts->sc, radiance, wl);
    if (tr) {
        memcpy(tr->wavelengths, wl, sizeof wl);
        tr->radiance[0] = radiance.x; tr->radiance[1] = radiance.y;
        tr->radiance[2] = radiance.z; tr->radiance[3] = radiance.w;
        tr->xyz[0] = xyz.x; tr->xyz[1] = xyz.y; tr->xyz[2] = xyz.z;
    }
    return xyz;
}

static int scene_ok(const orc_scene *sc)
{
    return sc && sc->primitives && sc->lights && sc->spectra && sc->cie && sc->camera &&
           sc->nlight > 0 && sc->nspectra > 0;
}

int orc_render(const orc_scene *sc, float *accum, uint8_t *rgba8, uint32_t first_sample,
               uint32_t n_samples, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               uint64_t counters[ORC_NCOUNTERS], int nthreads)
{
    if (!scene_ok(sc)) return -1;
    uint32_t W = (uint32_t)sc->camera[11], H = (uint32_t)sc->camera[12]; /* CS:85 */
    if (x1 > W) x1 = W;
    if (y1 > H) y1 = H;
    float frame[12];
    orc_camera_frame(sc->camera, frame);
    float hit_pad = orc_hit_pad(sc);
    uint64_t c_rays = 0, c_tests = 0, c_paths = 0, c_bounces = 0, c_shadow = 0, c_rand = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : c_rays, c_tests, c_paths, c_bounces, c_shadow, c_rand)
    for (int64_t yy = (int64_t)y0; yy < (int64_t)y1; yy++) {
        uint32_t y = (uint32_t)yy;
        tstate ts; memset(&ts, 0, sizeof ts);
        ts.sc = sc; ts.hit_pad = hit_pad;
        for (uint32_t x = x0; x < x1; x++) {
            size_t pix = (size_t)x + (size_t)y * W;          /* CS:107 */
            v3 acc = V3(0, 0, 0);
            if (accum) acc = V3(accum[pix * 4 + 0], accum[pix * 4 + 1], accum[pix * 4 + 2]);
            uint32_t last = first_sample;
            for (uint32_t s = 0; s < n_samples; s++) {
                ts.sample = first_sample + s;                /* UpdateVariables.wgsl: sample++ first */
                last = ts.sample;
                v3 xyz = pixel_sample(&ts, x, y, W, H, frame, NULL);
                acc = add3(acc, xyz);                        /* CS:108 */
                c_paths++;
            }
            if (accum) { accum[pix * 4 + 0] = acc.x; accum[pix * 4 + 1] = acc.y; accum[pix * 4 + 2] = acc.z; }
            if (rgba8 && n_samples > 0) {
                v3 avg = div3s(acc, (float)last);            /* CS:110 */
                v3 rgb = xyz_to_linear_rgb(avg);
                v3 ldr = tone_map(rgb, 2.2f);
                gamma_correct(&ldr);
                rgba8[pix * 4 + 0] = unorm8(ldr.x); rgba8[pix * 4 + 1] = unorm8(ldr.y);
                rgba8[pix * 4 + 2] = unorm8(ldr.z); rgba8[pix * 4 + 3] = 255;
            }
        }
        c_rays += ts.c_rays; c_tests += ts.c_tests; c_bounces += ts.c_bounces;
        c_shadow += ts.c_shadow; c_rand += ts.c_rand;
    }
    if (counters) {
        counters[0] += c_rays; counters[1] += c_tests; counters[2] += c_paths;
        counters[3] += c_bounces; counters[4] += c_shadow; counters[5] += c_rand;
    }
    return 0;
}

int orc_trace_pixel(const orc_scene *sc, uint32_t x, uint32_t y, uint32_t sample, orc_transcript *out)
{
    if (!scene_ok(sc) || !out) return -1;
    uint32_t W = (uint32_t)sc->camera[11], H = (uint32_t)sc->camera[12];
    float frame[12];
    orc_camera_frame(sc->camera, frame);
    tstate ts; memset(&ts, 0, sizeof ts);
    ts.sc = sc; ts.hit_pad = orc_hit_pad(sc); ts.sample = sample;
    memset(out, 0, sizeof *out);
    pixel_sample(&ts, x, y, W, H, frame, out);
    out->n_rand = (uint32_t)ts.c_rand;
    return 0;
}

/* Every intersect() call of one pixel-sample, in order: 12 floats each =
 * o.xyz, d.xyz, exclude bits, hit index bits (0xFFFFFFFF = miss), t, normal.xyz */
int orc_ray_log(const orc_scene *sc, uint32_t x, uint32_t y, uint32_t sample, float *log,
                uint32_t cap)
{
    if (!scene_ok(sc) || !log) return -1;
    uint32_t W = (uint32_t)sc->camera[11], H = (uint32_t)sc->camera[12];
    float frame[12];
    orc_camera_frame(sc->camera, frame);
    tstate ts; memset(&ts, 0, sizeof ts);
    ts.sc = sc; ts.hit_pad = orc_hit_pad(sc); ts.sample = sample;
    ts.raylog = log; ts.raylog_cap = cap;
    pixel_sample(&ts, x, y, W, H, frame, NULL);
    return (int)ts.raylog_n;
}

int orc_intersect(const orc_scene *sc, const float o[3], const float d[3], uint32_t exclude,
                  float out_f[7], uint32_t out_u[5])
{
    if (!sc || !sc->primitives) return -1;
    tstate ts; memset(&ts, 0, sizeof ts);
    ts.sc = sc; ts.hit_pad = orc_hit_pad(sc);
    ray_t r = { V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]) };
    isect_t is = intersect(&ts, &r, exclude);
    out_f[0] = is.ctx.t_max;
    out_f[1] = is.si.position.x; out_f[2] = is.si.position.y; out_f[3] = is.si.position.z;
    out_f[4] = is.si.normal.x; out_f[5] = is.si.normal.y; out_f[6] = is.si.normal.z;
    out_u[0] = (uint32_t)is.ctx.hit; out_u[1] = is.ctx.index; out_u[2] = is.si.material;
    out_u[3] = is.si.emission_index; out_u[4] = is.si.reflectance_index;
    return 0;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

const char *orc_version(void) { return "crt-oracle 1 (brute force; ComputeShader.wgsl restatement)"; }
