/*
 * crt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, brute force, bug-compatible) of the reference's
 * path-trace compute pass:
 *     /root/reference/src/shaders/ComputeShader.wgsl   (live set)
 *     /root/reference/src/shaders/UpdateVariables.wgsl (sample++)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (computeraytracer_amd/, include/, addon/, host/)
 * never includes, links or calls anything in this directory.
 *
 * PARITY PINNING: the reference ships no tests, fixtures or golden images
 * (package.json:9) and its WGSL cannot be executed in the build container
 * (no WebGPU runtime), so float results are "parity unpinned" against a real
 * WGSL stack.  What IS pinned, implementation-independently: the integer RNG
 * (tea/pcg4d KATs), the packed buffer sizes and the SHA-256 of
 * the spectra / CIE tables.  See tests/test_oracle_kat.py.
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Host buffers exactly as src/main.js packs them (bind group 0, b4..b8). */
typedef struct {
    const uint8_t *primitives; /* nprim  x 80 B  (main.js:211-246)            */
    uint32_t nprim;
    const uint8_t *lights;     /* nlight x 80 B  (main.js:255-296)            */
    uint32_t nlight;
    const float *spectra;      /* nspectra x 301 (main.js:334-378)            */
    uint32_t nspectra;
    const float *cie;          /* 3 x 471        (main.js:380-393)            */
    const float *camera;       /* 16 floats      (main.js:313-324)            */
} orc_scene;

/* counters[]: 0 rays (= intersect() invocations), 1 primitive tests,
 * 2 paths, 3 path-loop iterations (bounces), 4 shadow rays, 5 rand() calls */
enum { ORC_NCOUNTERS = 8 };

/* Renders samples first_sample .. first_sample+n_samples-1 (the value the
 * kernel sees in `sample`; the first frame after reset sees 1) for pixels
 * x0<=x<x1, y0<=y<y1 of the full W x H image named by camera[11], camera[12].
 * accum: full-image W*H*4 floats (xyz + pad, stride 16 B like
 * alt_color_buffer), read-modify-written.  rgba8: full-image W*H*4 bytes,
 * row 0 = top.  Either may be NULL.  nthreads<=0 -> all cores (OpenMP).      */
int orc_render(const orc_scene *sc, float *accum, uint8_t *rgba8,
               uint32_t first_sample, uint32_t n_samples,
               uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               uint64_t counters[ORC_NCOUNTERS], int nthreads);

/* Per-path transcript for probe pixels: sequence of closest-hit indices of
 * the path loop (0xFFFFFFFF = miss), number of rand() calls, radiance. */
typedef struct {
    uint32_t n_hits;
    uint32_t hits[128];
    uint32_t n_rand;
    uint32_t wavelengths[4];
    float radiance[4];
    float xyz[3];
} orc_transcript;
int orc_trace_pixel(const orc_scene *sc, uint32_t x, uint32_t y, uint32_t sample,
                    orc_transcript *out);

/* Every intersect() call of one pixel-sample, in order; 12 floats per ray:
 * o.xyz, d.xyz, exclude bits, hit index bits (0xFFFFFFFF = miss), t, normal.xyz.
 * Returns the number of rays logged (<= cap) or -1. */
int orc_ray_log(const orc_scene *sc, uint32_t x, uint32_t y, uint32_t sample, float *log,
                uint32_t cap);

/* One closest-hit query with the reference's brute-force loop (for BVH tests).
 * out_f: t, px,py,pz, nx,ny,nz ; out_u: hit, index, material, emission, reflectance */
int orc_intersect(const orc_scene *sc, const float o[3], const float d[3],
                  uint32_t exclude, float out_f[7], uint32_t out_u[5]);

/* Integer-exact RNG pieces (ComputeShader.wgsl:865-897). */
uint32_t orc_tea(uint32_t v0, uint32_t v1);
/* seeds (y, x*100, sample, tea(x, y*100)), then n x rand(); writes the 24-bit
 * integers seed.x & 0xFFFFFF and the final seed. */
void orc_rand_kat(uint32_t x, uint32_t y, uint32_t sample, uint32_t n,
                  uint32_t *out24, uint32_t seed_out[4]);

/* Deterministic f32 math spec (elementwise, for pinning the device math). */
void orc_math_eval(int fn, const float *a, const float *b, float *out, size_t n);
enum { ORC_FN_SIN = 0, ORC_FN_COS = 1, ORC_FN_EXP = 2, ORC_FN_LOG2 = 3,
       ORC_FN_EXP2 = 4, ORC_FN_POW = 5, ORC_FN_SQRT = 6, ORC_FN_DIV = 7,
       ORC_FN_TAN = 8 };

/* Scene-scale hit pad used by the category-2 (triangle) acceptance rule. */
float orc_hit_pad(const orc_scene *sc);

/* Camera frame as hoisted to the host: 12 floats llc, horizontal, vertical, eye */
void orc_camera_frame(const float camera[16], float out[12]);

/* Threads orc_render uses when nthreads <= 0. */
int orc_max_threads(void);

const char *orc_version(void);

#ifdef __cplusplus
}
#endif
#endif
