/*
 * crt_napi.c -- thin N-API addon: 1:1 JavaScript wrappers over the C ABI of
 * include/crt.h (libcrt.so, hand-written HIP for gfx950).
 *
 * It stands where the reference's src/main.js talks to WebGPU:
 *   navigator.gpu.requestAdapter/requestDevice (main.js:8-9)        -> create()
 *   createBuffer + getMappedRange + unmap for b4..b8 (main.js:147-393) -> uploadScene()
 *   dispatchWorkgroups(1); dispatchWorkgroups(W/8,H/8) (main.js:598-611) -> trace(n)
 *   "uncapturederror" (main.js:11-14)                               -> thrown Error
 * Build: plain gcc against /usr/include/node (no node-gyp), see addon/Makefile.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/crt.h"

#define NAPI_OK(env, call)                                                          \
    do {                                                                            \
        if ((call) != napi_ok) {                                                    \
            napi_throw_error((env), NULL, "N-API call failed: " #call);             \
            return NULL;                                                            \
        }                                                                           \
    } while (0)

static napi_value throw_crt(napi_env env, crt_ctx *ctx, int code, const char *what)
{
    char msg[640];
    const char *detail = crt_last_error(ctx);
    snprintf(msg, sizeof msg, "%s failed (%d): %s", what, code, detail ? detail : "");
    napi_throw_error(env, "ERR_CRT", msg);
    return NULL;
}

#define CRT_CHECK(env, ctx, what, call)                                  \
    do {                                                                 \
        int rc_ = (call);                                                \
        if (rc_ != CRT_OK) return throw_crt((env), (ctx), rc_, (what));  \
    } while (0)

/* What a handle holds: the context and the queue of asynchronous jobs on it.  The C ABI allows one thread per
 * context at a time, so the *Async entry points run their jobs one after the other (napi_async_work on the libuv
 * pool, the next one queued when the previous completes) and the synchronous entry points refuse to run while
 * jobs are pending ("await the promises first"). */
struct job;
typedef struct slot {
    crt_ctx *ctx;
    struct job *head, *tail;     /* pending jobs, head = the one running */
    napi_ref self;               /* strong reference to the handle while jobs are pending: a promise does not keep the
                                    external alive, and `traceAsync(create(0), n)` must not be collected mid-job */
} slot;

static void finalize_ctx(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    slot *s = (slot *)data;
    if (s) {
        /* (unreachable with jobs pending: `self` holds the handle until the queue is empty) */
        if (s->ctx) crt_destroy(s->ctx);
        free(s);
    }
}

static slot *get_slot(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((slot *)p)->ctx) {
        napi_throw_type_error(env, NULL, "invalid or destroyed crt context handle");
        return NULL;
    }
    return (slot *)p;
}

/* args[0] is always the handle (an external holding a slot); synchronous calls need an idle context */
static crt_ctx *get_ctx(napi_env env, napi_value v)
{
    slot *s = get_slot(env, v);
    if (!s) return NULL;
    if (s->head) {
        napi_throw_error(env, "ERR_CRT_BUSY", "asynchronous calls are pending on this context: await their promises first");
        return NULL;
    }
    return s->ctx;
}

/* Bytes of an ArrayBuffer / TypedArray / DataView / Buffer argument. */
static int get_bytes(napi_env env, napi_value v, void **data, size_t *len)
{
    bool is;
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is)
        return napi_get_arraybuffer_info(env, v, data, len) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type t; size_t n, off; napi_value ab;
        if (napi_get_typedarray_info(env, v, &t, &n, data, &ab, &off) != napi_ok) return 0;
        static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        if ((size_t)t >= sizeof esz / sizeof esz[0]) return 0;      /* (an element type this table does not know) */
        *len = n * esz[t];
        return 1;
    }
    if (napi_is_dataview(env, v, &is) == napi_ok && is) {
        napi_value ab; size_t off;
        return napi_get_dataview_info(env, v, len, data, &ab, &off) == napi_ok;
    }
    if (napi_is_buffer(env, v, &is) == napi_ok && is)
        return napi_get_buffer_info(env, v, data, len) == napi_ok;
    return 0;
}

#define ARGS(n)                                                       \
    size_t argc = (n);                                                \
    napi_value argv[(n) > 0 ? (n) : 1];                               \
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL)); \
    if (argc < (size_t)(n)) { napi_throw_type_error(env, NULL, "too few arguments"); return NULL; }

static napi_value undefined(napi_env env)
{
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

static napi_value js_create(napi_env env, napi_callback_info info)
{
    ARGS(1)
    int32_t dev = 0;
    NAPI_OK(env, napi_get_value_int32(env, argv[0], &dev));
    crt_ctx *ctx = NULL;
    int rc = crt_create(&ctx, dev);
    if (rc != CRT_OK) return throw_crt(env, NULL, rc, "crt_create");
    slot *sl = (slot *)calloc(1, sizeof *sl);
    sl->ctx = ctx;
    napi_value ext;
    NAPI_OK(env, napi_create_external(env, sl, finalize_ctx, NULL, &ext));
    return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info)
{
    ARGS(1)
    void *p = NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p && ((slot *)p)->ctx) {
        if (((slot *)p)->head) {
            napi_throw_error(env, "ERR_CRT_BUSY", "destroy: asynchronous calls are pending on this context");
            return NULL;
        }
        crt_destroy(((slot *)p)->ctx);
        ((slot *)p)->ctx = NULL;
    }
    return undefined(env);
}

/* uploadScene(h, primitives, lights, spectra, cie, camera) -- byte buffers exactly as
 * main.js builds them (80-byte records; Float32Array tables). */
static napi_value js_upload_scene(napi_env env, napi_callback_info info)
{
    ARGS(6)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p[5]; size_t n[5];
    static const char *names[5] = {"primitives", "lights", "spectra", "cie", "camera"};
    for (int i = 0; i < 5; i++) {
        if (!get_bytes(env, argv[i + 1], &p[i], &n[i])) {
            char m[96];
            snprintf(m, sizeof m, "uploadScene: %s must be an ArrayBuffer or typed array", names[i]);
            napi_throw_type_error(env, NULL, m);
            return NULL;
        }
    }
    if (n[0] % 80 || n[1] % 80 || n[2] % (301 * 4) || n[3] != 3 * 471 * 4 || n[4] != 64) {
        napi_throw_range_error(env, NULL, "uploadScene: sizes must be primitives/lights k*80 B, spectra k*1204 B, "
                                          "cie 5652 B, camera 64 B");
        return NULL;
    }
    CRT_CHECK(env, ctx, "crt_upload_scene",
              crt_upload_scene(ctx, p[0], n[0] / 80, p[1], n[1] / 80, (const float *)p[2], n[2] / (301 * 4),
                               (const float *)p[3], (const float *)p[4]));
    return undefined(env);
}

static napi_value js_set_tile(napi_env env, napi_callback_info info)
{
    ARGS(5)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t v[4];
    for (int i = 0; i < 4; i++) NAPI_OK(env, napi_get_value_uint32(env, argv[i + 1], &v[i]));
    CRT_CHECK(env, ctx, "crt_set_tile", crt_set_tile(ctx, v[0], v[1], v[2], v[3]));
    return undefined(env);
}

static napi_value js_build_accel(napi_env env, napi_callback_info info)
{
    ARGS(2)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int32_t mode;
    NAPI_OK(env, napi_get_value_int32(env, argv[1], &mode));
    CRT_CHECK(env, ctx, "crt_build_accel", crt_build_accel(ctx, mode));
    return undefined(env);
}

static napi_value js_reset(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    CRT_CHECK(env, ctx, "crt_reset", crt_reset(ctx));
    return undefined(env);
}

static napi_value js_trace(napi_env env, napi_callback_info info)
{
    ARGS(2)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t n;
    NAPI_OK(env, napi_get_value_uint32(env, argv[1], &n));
    CRT_CHECK(env, ctx, "crt_trace", crt_trace(ctx, n));
    return undefined(env);
}

static napi_value js_sync(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    CRT_CHECK(env, ctx, "crt_sync", crt_sync(ctx));
    return undefined(env);
}

static napi_value js_sample_count(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t s = 0;
    CRT_CHECK(env, ctx, "crt_sample_count", crt_sample_count(ctx, &s));
    napi_value out;
    NAPI_OK(env, napi_create_uint32(env, s, &out));
    return out;
}

static napi_value js_tile(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t t[4];
    CRT_CHECK(env, ctx, "crt_tile", crt_tile(ctx, t));
    napi_value arr;
    NAPI_OK(env, napi_create_array_with_length(env, 4, &arr));
    for (uint32_t i = 0; i < 4; i++) {
        napi_value v;
        NAPI_OK(env, napi_create_uint32(env, t[i], &v));
        NAPI_OK(env, napi_set_element(env, arr, i, v));
    }
    return arr;
}

static napi_value read_image(napi_env env, napi_callback_info info, int rgba)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t t[4];
    CRT_CHECK(env, ctx, "crt_tile", crt_tile(ctx, t));
    size_t px = (size_t)t[2] * t[3];
    size_t bytes = px * (rgba ? 4 : 16);
    void *data = NULL;
    napi_value ab, ta;
    NAPI_OK(env, napi_create_arraybuffer(env, bytes, &data, &ab));
    if (rgba) CRT_CHECK(env, ctx, "crt_read_rgba8", crt_read_rgba8(ctx, (uint8_t *)data));
    else CRT_CHECK(env, ctx, "crt_read_accum", crt_read_accum(ctx, (float *)data));
    NAPI_OK(env, napi_create_typedarray(env, rgba ? napi_uint8_array : napi_float32_array, px * 4, ab, 0, &ta));
    return ta;
}
static napi_value js_read_accum(napi_env env, napi_callback_info info) { return read_image(env, info, 0); }
static napi_value js_read_rgba8(napi_env env, napi_callback_info info) { return read_image(env, info, 1); }

static napi_value js_write_accum(napi_env env, napi_callback_info info)
{
    ARGS(3)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p; size_t n;
    uint32_t s, t[4];
    if (!get_bytes(env, argv[1], &p, &n)) { napi_throw_type_error(env, NULL, "writeAccum: typed array expected"); return NULL; }
    NAPI_OK(env, napi_get_value_uint32(env, argv[2], &s));
    CRT_CHECK(env, ctx, "crt_tile", crt_tile(ctx, t));
    if (n != (size_t)t[2] * t[3] * 16) { napi_throw_range_error(env, NULL, "writeAccum: need tw*th*4 floats"); return NULL; }
    CRT_CHECK(env, ctx, "crt_write_accum", crt_write_accum(ctx, (const float *)p, s));
    return undefined(env);
}

static napi_value js_enable_counters(napi_env env, napi_callback_info info)
{
    ARGS(2)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    bool on;
    NAPI_OK(env, napi_get_value_bool(env, argv[1], &on));
    CRT_CHECK(env, ctx, "crt_enable_counters", crt_enable_counters(ctx, on ? 1 : 0));
    return undefined(env);
}

static napi_value js_reset_counters(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    CRT_CHECK(env, ctx, "crt_reset_counters", crt_reset_counters(ctx));
    return undefined(env);
}

static napi_value u64_array(napi_env env, const uint64_t *v, uint32_t n)
{
    napi_value arr;
    NAPI_OK(env, napi_create_array_with_length(env, n, &arr));
    for (uint32_t i = 0; i < n; i++) {
        napi_value d;
        NAPI_OK(env, napi_create_double(env, (double)v[i], &d));   /* exact below 2^53 */
        NAPI_OK(env, napi_set_element(env, arr, i, d));
    }
    return arr;
}

static napi_value js_counters(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint64_t c[CRT_NCOUNTERS];
    CRT_CHECK(env, ctx, "crt_counters", crt_counters(ctx, c));
    return u64_array(env, c, CRT_NCOUNTERS);
}

static napi_value js_accel_stats(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint64_t c[8];
    CRT_CHECK(env, ctx, "crt_accel_stats", crt_accel_stats(ctx, c));
    return u64_array(env, c, 8);
}

static napi_value js_last_trace_ms(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    float ms = 0; uint32_t n = 0;
    CRT_CHECK(env, ctx, "crt_last_trace_ms", crt_last_trace_ms(ctx, &ms, &n));
    napi_value arr, a, b;
    NAPI_OK(env, napi_create_array_with_length(env, 2, &arr));
    NAPI_OK(env, napi_create_double(env, ms, &a));
    NAPI_OK(env, napi_create_uint32(env, n, &b));
    NAPI_OK(env, napi_set_element(env, arr, 0, a));
    NAPI_OK(env, napi_set_element(env, arr, 1, b));
    return arr;
}

static napi_value js_set_option(napi_env env, napi_callback_info info)
{
    ARGS(3)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    char name[64]; size_t len; int64_t v;
    NAPI_OK(env, napi_get_value_string_utf8(env, argv[1], name, sizeof name, &len));
    NAPI_OK(env, napi_get_value_int64(env, argv[2], &v));
    CRT_CHECK(env, ctx, "crt_set_option", crt_set_option(ctx, name, v));
    return undefined(env);
}

/* The display step without stopping the pipeline (include/crt.h): the newest complete frame + its sample index. */
static napi_value js_read_latest_rgba8(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t t[4], smp = 0;
    CRT_CHECK(env, ctx, "crt_tile", crt_tile(ctx, t));
    size_t px = (size_t)t[2] * t[3];
    void *data = NULL;
    napi_value ab, ta, obj, sv;
    NAPI_OK(env, napi_create_arraybuffer(env, px * 4, &data, &ab));
    CRT_CHECK(env, ctx, "crt_read_latest_rgba8", crt_read_latest_rgba8(ctx, (uint8_t *)data, &smp));
    NAPI_OK(env, napi_create_typedarray(env, napi_uint8_array, px * 4, ab, 0, &ta));
    NAPI_OK(env, napi_create_object(env, &obj));
    NAPI_OK(env, napi_create_uint32(env, smp, &sv));
    NAPI_OK(env, napi_set_named_property(env, obj, "frame", ta));
    NAPI_OK(env, napi_set_named_property(env, obj, "sample", sv));
    return obj;
}

static napi_value js_latest_sample(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t s = 0;
    CRT_CHECK(env, ctx, "crt_latest_sample", crt_latest_sample(ctx, &s));
    napi_value out;
    NAPI_OK(env, napi_create_uint32(env, s, &out));
    return out;
}

static napi_value js_read_sample_rgba8(napi_env env, napi_callback_info info)
{
    ARGS(2)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t t[4], smp;
    NAPI_OK(env, napi_get_value_uint32(env, argv[1], &smp));
    CRT_CHECK(env, ctx, "crt_tile", crt_tile(ctx, t));
    size_t px = (size_t)t[2] * t[3];
    void *data = NULL;
    napi_value ab, ta;
    NAPI_OK(env, napi_create_arraybuffer(env, px * 4, &data, &ab));
    CRT_CHECK(env, ctx, "crt_read_sample_rgba8", crt_read_sample_rgba8(ctx, smp, (uint8_t *)data));
    NAPI_OK(env, napi_create_typedarray(env, napi_uint8_array, px * 4, ab, 0, &ta));
    return ta;
}

/* pinHost(typedArray) / unpinHost(typedArray): page-lock the array's memory; readSampleRgba8Into(h, k, typedArray) then reads
 * frame k straight into it (the caller's frame buffer, reused frame after frame). */
static napi_value js_pin_host(napi_env env, napi_callback_info info)
{
    ARGS(1)
    void *p; size_t n;
    if (!get_bytes(env, argv[0], &p, &n)) { napi_throw_type_error(env, NULL, "pinHost: typed array expected"); return NULL; }
    int rc = crt_pin_host(p, n);
    if (rc != CRT_OK) return throw_crt(env, NULL, rc, "crt_pin_host");
    return undefined(env);
}

static napi_value js_unpin_host(napi_env env, napi_callback_info info)
{
    ARGS(1)
    void *p; size_t n;
    if (!get_bytes(env, argv[0], &p, &n)) { napi_throw_type_error(env, NULL, "unpinHost: typed array expected"); return NULL; }
    int rc = crt_unpin_host(p);
    if (rc != CRT_OK) return throw_crt(env, NULL, rc, "crt_unpin_host");
    return undefined(env);
}

static napi_value js_read_sample_rgba8_into(napi_env env, napi_callback_info info)
{
    ARGS(3)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t t[4], smp;
    void *p; size_t n;
    NAPI_OK(env, napi_get_value_uint32(env, argv[1], &smp));
    if (!get_bytes(env, argv[2], &p, &n)) { napi_throw_type_error(env, NULL, "readSampleRgba8Into: typed array expected"); return NULL; }
    CRT_CHECK(env, ctx, "crt_tile", crt_tile(ctx, t));
    if (n < (size_t)t[2] * t[3] * 4) { napi_throw_range_error(env, NULL, "readSampleRgba8Into: need tw*th*4 bytes"); return NULL; }
    CRT_CHECK(env, ctx, "crt_read_sample_rgba8", crt_read_sample_rgba8(ctx, smp, (uint8_t *)p));
    return undefined(env);
}

/* ------------------------------------------------------------------ multi-GPU (include/crt.h "Multi-GPU") and composition
 * The reference drives one GPUDevice (src/main.js:8-9); a Node host reaches the tile-partitioned configurations through
 * these: one process (worker) per GPU, the communicator id made by one of them and passed around by the parent
 * (host/multi.js). */
static napi_value js_comm_unique_id(napi_env env, napi_callback_info info)
{
    ARGS(1)
    bool local = false;
    NAPI_OK(env, napi_get_value_bool(env, argv[0], &local));
    void *data = NULL;
    napi_value ab;
    NAPI_OK(env, napi_create_arraybuffer(env, CRT_COMM_ID_BYTES, &data, &ab));
    int rc = crt_comm_unique_id(data, local ? CRT_COMM_LOCAL : CRT_COMM_RCCL);
    if (rc != CRT_OK) return throw_crt(env, NULL, rc, "crt_comm_unique_id");
    return ab;
}

static napi_value js_comm_init(napi_env env, napi_callback_info info)
{
    ARGS(4)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p; size_t n;
    int32_t rank, world;
    if (!get_bytes(env, argv[1], &p, &n) || n != CRT_COMM_ID_BYTES) { napi_throw_type_error(env, NULL, "commInit: the id is the 128-byte buffer of commUniqueId"); return NULL; }
    NAPI_OK(env, napi_get_value_int32(env, argv[2], &rank));
    NAPI_OK(env, napi_get_value_int32(env, argv[3], &world));
    CRT_CHECK(env, ctx, "crt_comm_init", crt_comm_init(ctx, p, rank, world));
    return undefined(env);
}

static napi_value js_comm_partition(napi_env env, napi_callback_info info)
{
    ARGS(2)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t band;
    NAPI_OK(env, napi_get_value_uint32(env, argv[1], &band));
    CRT_CHECK(env, ctx, "crt_comm_partition", crt_comm_partition(ctx, band));
    return undefined(env);
}

static napi_value js_comm_info(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int v[4];
    CRT_CHECK(env, ctx, "crt_comm_info", crt_comm_info(ctx, v));
    napi_value arr;
    NAPI_OK(env, napi_create_array_with_length(env, 4, &arr));
    for (uint32_t i = 0; i < 4; i++) {
        napi_value e;
        NAPI_OK(env, napi_create_int32(env, v[i], &e));
        NAPI_OK(env, napi_set_element(env, arr, i, e));
    }
    return arr;
}

static napi_value js_comm_destroy(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    CRT_CHECK(env, ctx, "crt_comm_destroy", crt_comm_destroy(ctx));
    return undefined(env);
}

static napi_value js_gather(napi_env env, napi_callback_info info)
{
    ARGS(2)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int32_t what;
    NAPI_OK(env, napi_get_value_int32(env, argv[1], &what));
    CRT_CHECK(env, ctx, "crt_gather", crt_gather(ctx, what));
    return undefined(env);
}

static napi_value read_frame(napi_env env, napi_callback_info info, int rgba)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t wh[2];
    CRT_CHECK(env, ctx, "crt_image_size", crt_image_size(ctx, wh));
    size_t px = (size_t)wh[0] * wh[1];
    void *data = NULL;
    napi_value ab, ta;
    NAPI_OK(env, napi_create_arraybuffer(env, px * (rgba ? 4 : 16), &data, &ab));
    if (rgba) CRT_CHECK(env, ctx, "crt_read_frame_rgba8", crt_read_frame_rgba8(ctx, (uint8_t *)data));
    else CRT_CHECK(env, ctx, "crt_read_frame_accum", crt_read_frame_accum(ctx, (float *)data));
    NAPI_OK(env, napi_create_typedarray(env, rgba ? napi_uint8_array : napi_float32_array, px * 4, ab, 0, &ta));
    return ta;
}
static napi_value js_read_frame_rgba8(napi_env env, napi_callback_info info) { return read_frame(env, info, 1); }
static napi_value js_read_frame_accum(napi_env env, napi_callback_info info) { return read_frame(env, info, 0); }

static napi_value js_image_size(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t wh[2];
    CRT_CHECK(env, ctx, "crt_image_size", crt_image_size(ctx, wh));
    napi_value arr, a, b;
    NAPI_OK(env, napi_create_array_with_length(env, 2, &arr));
    NAPI_OK(env, napi_create_uint32(env, wh[0], &a));
    NAPI_OK(env, napi_create_uint32(env, wh[1], &b));
    NAPI_OK(env, napi_set_element(env, arr, 0, a));
    NAPI_OK(env, napi_set_element(env, arr, 1, b));
    return arr;
}

static napi_value js_set_row_bands(napi_env env, napi_callback_info info)
{
    ARGS(4)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t v[3];
    for (int i = 0; i < 3; i++) NAPI_OK(env, napi_get_value_uint32(env, argv[i + 1], &v[i]));
    CRT_CHECK(env, ctx, "crt_set_row_bands", crt_set_row_bands(ctx, v[0], v[1], v[2]));
    return undefined(env);
}

/* Device addresses cross the boundary as BigInt (or a Number / null for "none"). */
static int get_ptr(napi_env env, napi_value v, void **out)
{
    napi_valuetype t;
    *out = NULL;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_null || t == napi_undefined) return 1;
    if (t == napi_bigint) { uint64_t u; bool lossless; if (napi_get_value_bigint_uint64(env, v, &u, &lossless) != napi_ok) return 0; *out = (void *)(uintptr_t)u; return 1; }
    if (t == napi_number) { int64_t i; if (napi_get_value_int64(env, v, &i) != napi_ok || i < 0) return 0; *out = (void *)(uintptr_t)i; return 1; }
    return 0;
}

static napi_value ptr_pair(napi_env env, void *a, void *b)
{
    napi_value arr, x, y;
    NAPI_OK(env, napi_create_array_with_length(env, 2, &arr));
    NAPI_OK(env, napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)a, &x));
    NAPI_OK(env, napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)b, &y));
    NAPI_OK(env, napi_set_element(env, arr, 0, x));
    NAPI_OK(env, napi_set_element(env, arr, 1, y));
    return arr;
}

static napi_value js_device_buffers(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *a = NULL, *b = NULL;
    CRT_CHECK(env, ctx, "crt_device_buffers", crt_device_buffers(ctx, &a, &b));
    return ptr_pair(env, a, b);
}

static napi_value js_frame_device_buffers(napi_env env, napi_callback_info info)
{
    ARGS(1)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *a = NULL, *b = NULL;
    CRT_CHECK(env, ctx, "crt_frame_device_buffers", crt_frame_device_buffers(ctx, &a, &b));
    return ptr_pair(env, a, b);
}

static napi_value js_bind_output(napi_env env, napi_callback_info info)
{
    ARGS(3)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *a, *b;
    if (!get_ptr(env, argv[1], &a) || !get_ptr(env, argv[2], &b)) { napi_throw_type_error(env, NULL, "bindOutput: device addresses are BigInt (or null)"); return NULL; }
    CRT_CHECK(env, ctx, "crt_bind_output", crt_bind_output(ctx, a, b));
    return undefined(env);
}

static napi_value js_set_stream(napi_env env, napi_callback_info info)
{
    ARGS(2)
    crt_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *st;
    if (!get_ptr(env, argv[1], &st)) { napi_throw_type_error(env, NULL, "setStream: a hipStream_t as BigInt (or null for the context's own)"); return NULL; }
    CRT_CHECK(env, ctx, "crt_set_stream", crt_set_stream(ctx, st));
    return undefined(env);
}

/* ------------------------------------------------------------------ asynchronous entry points
 * traceAsync(h, n), syncAsync(h), readRgba8Async(h), readAccumAsync(h) -> Promise.  The reference's frame() is
 * fire-and-forget (queue.submit, src/main.js:618-620): a Node display loop must not block its event loop on the
 * GPU either.  Jobs of one context run in call order. */
enum { JOB_TRACE, JOB_SYNC, JOB_READ_RGBA8, JOB_READ_ACCUM, JOB_GATHER, JOB_READ_FRAME_RGBA8, JOB_READ_FRAME_ACCUM, JOB_READ_SAMPLE_RGBA8 };
typedef struct job {
    napi_async_work work;
    napi_deferred deferred;
    slot *sl;
    int op, rc;
    uint32_t n, px;
    void *data;              /* ArrayBuffer memory of a read job (kept alive by ab_ref) */
    napi_ref ab_ref;
    char err[640];
    struct job *next;
} job;

static void job_execute(napi_env env, void *data)
{
    (void)env;
    job *j = (job *)data;
    crt_ctx *ctx = j->sl->ctx;
    switch (j->op) {
    case JOB_TRACE: j->rc = crt_trace(ctx, j->n); break;
    case JOB_SYNC: j->rc = crt_sync(ctx); break;
    case JOB_READ_RGBA8: j->rc = crt_read_rgba8(ctx, (uint8_t *)j->data); break;
    case JOB_READ_ACCUM: j->rc = crt_read_accum(ctx, (float *)j->data); break;
    case JOB_GATHER: j->rc = crt_gather(ctx, (int)j->n); break;
    case JOB_READ_FRAME_RGBA8: j->rc = crt_read_frame_rgba8(ctx, (uint8_t *)j->data); break;
    case JOB_READ_SAMPLE_RGBA8: j->rc = crt_read_sample_rgba8(ctx, j->n, (uint8_t *)j->data); break;
    default: j->rc = crt_read_frame_accum(ctx, (float *)j->data); break;
    }
    if (j->rc != CRT_OK) {
        const char *d = crt_last_error(ctx);
        snprintf(j->err, sizeof j->err, "asynchronous call failed (%d): %s", j->rc, d ? d : "");
    }
}

static void job_complete(napi_env env, napi_status status, void *data)
{
    job *j = (job *)data;
    slot *sl = j->sl;
    napi_value result = NULL;
    if (status != napi_ok && j->rc == CRT_OK) { j->rc = CRT_EDEVICE; snprintf(j->err, sizeof j->err, "asynchronous call cancelled"); }
    if (j->rc == CRT_OK) {
        if (j->ab_ref) {
            napi_value ab;
            const int bytes8 = j->op == JOB_READ_RGBA8 || j->op == JOB_READ_FRAME_RGBA8 || j->op == JOB_READ_SAMPLE_RGBA8;
            if (napi_get_reference_value(env, j->ab_ref, &ab) == napi_ok)
                napi_create_typedarray(env, bytes8 ? napi_uint8_array : napi_float32_array, (size_t)j->px * 4, ab, 0, &result);
        }
        if (!result) napi_get_undefined(env, &result);
        napi_resolve_deferred(env, j->deferred, result);
    } else {
        napi_value msg, code, errv;
        napi_create_string_utf8(env, j->err, NAPI_AUTO_LENGTH, &msg);
        napi_create_string_utf8(env, "ERR_CRT", NAPI_AUTO_LENGTH, &code);
        napi_create_error(env, code, msg, &errv);
        napi_reject_deferred(env, j->deferred, errv);
    }
    if (j->ab_ref) napi_delete_reference(env, j->ab_ref);
    napi_delete_async_work(env, j->work);
    /* the next job of this context */
    sl->head = j->next;
    if (!sl->head) {
        sl->tail = NULL;
        if (sl->self) { napi_ref r = sl->self; sl->self = NULL; napi_delete_reference(env, r); }   /* the handle may go now */
    } else napi_queue_async_work(env, sl->head->work);
    free(j);
}

static napi_value start_job(napi_env env, napi_callback_info info, int op)
{
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < ((op == JOB_TRACE || op == JOB_GATHER || op == JOB_READ_SAMPLE_RGBA8) ? 2u : 1u)) { napi_throw_type_error(env, NULL, "too few arguments"); return NULL; }
    slot *sl = get_slot(env, argv[0]);
    if (!sl) return NULL;
    job *j = (job *)calloc(1, sizeof *j);
    j->sl = sl; j->op = op;
    if (op == JOB_TRACE || op == JOB_GATHER || op == JOB_READ_SAMPLE_RGBA8) {
        if (napi_get_value_uint32(env, argv[1], &j->n) != napi_ok) { free(j); napi_throw_type_error(env, NULL, "traceAsync / gatherAsync: a number expected"); return NULL; }
    }
    if (op == JOB_READ_RGBA8 || op == JOB_READ_ACCUM || op == JOB_READ_FRAME_RGBA8 || op == JOB_READ_FRAME_ACCUM || op == JOB_READ_SAMPLE_RGBA8) {
        const int frame = op == JOB_READ_FRAME_RGBA8 || op == JOB_READ_FRAME_ACCUM, bytes8 = op == JOB_READ_RGBA8 || op == JOB_READ_FRAME_RGBA8 || op == JOB_READ_SAMPLE_RGBA8;
        uint32_t t[4];
        if (frame) { if (crt_image_size(sl->ctx, t + 2) != CRT_OK) { free(j); return throw_crt(env, sl->ctx, CRT_ESTATE, "crt_image_size"); } }
        else if (crt_tile(sl->ctx, t) != CRT_OK) { free(j); return throw_crt(env, sl->ctx, CRT_ESTATE, "crt_tile"); }
        j->px = t[2] * t[3];
        napi_value ab;
        if (napi_create_arraybuffer(env, (size_t)j->px * (bytes8 ? 4 : 16), &j->data, &ab) != napi_ok ||
            napi_create_reference(env, ab, 1, &j->ab_ref) != napi_ok) { free(j); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    }
    napi_value promise, name;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
        napi_create_string_utf8(env, "crt_async", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, job_execute, job_complete, j, &j->work) != napi_ok) {
        free(j);
        napi_throw_error(env, NULL, "crt_napi: could not create the asynchronous job");
        return NULL;
    }
    if (sl->tail) { sl->tail->next = j; sl->tail = j; }
    else {
        if (!sl->self && napi_create_reference(env, argv[0], 1, &sl->self) != napi_ok) {
            sl->self = NULL;
            napi_delete_async_work(env, j->work);
            if (j->ab_ref) napi_delete_reference(env, j->ab_ref);
            free(j);
            napi_throw_error(env, NULL, "crt_napi: could not pin the context handle");
            return NULL;
        }
        sl->head = sl->tail = j;
        napi_queue_async_work(env, j->work);
    }
    return promise;
}
static napi_value js_trace_async(napi_env env, napi_callback_info info) { return start_job(env, info, JOB_TRACE); }
static napi_value js_sync_async(napi_env env, napi_callback_info info) { return start_job(env, info, JOB_SYNC); }
static napi_value js_read_rgba8_async(napi_env env, napi_callback_info info) { return start_job(env, info, JOB_READ_RGBA8); }
static napi_value js_read_accum_async(napi_env env, napi_callback_info info) { return start_job(env, info, JOB_READ_ACCUM); }
static napi_value js_gather_async(napi_env env, napi_callback_info info) { return start_job(env, info, JOB_GATHER); }
static napi_value js_read_sample_rgba8_async(napi_env env, napi_callback_info info) { return start_job(env, info, JOB_READ_SAMPLE_RGBA8); }
static napi_value js_read_frame_rgba8_async(napi_env env, napi_callback_info info) { return start_job(env, info, JOB_READ_FRAME_RGBA8); }
static napi_value js_read_frame_accum_async(napi_env env, napi_callback_info info) { return start_job(env, info, JOB_READ_FRAME_ACCUM); }

static napi_value js_abi_version(napi_env env, napi_callback_info info)
{
    (void)info;
    napi_value v;
    NAPI_OK(env, napi_create_int32(env, crt_abi_version(), &v));
    return v;
}

static napi_value init(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"create", js_create}, {"destroy", js_destroy}, {"uploadScene", js_upload_scene},
        {"setTile", js_set_tile}, {"buildAccel", js_build_accel}, {"reset", js_reset},
        {"trace", js_trace}, {"sync", js_sync}, {"sampleCount", js_sample_count}, {"tile", js_tile},
        {"readAccum", js_read_accum}, {"readRgba8", js_read_rgba8}, {"writeAccum", js_write_accum},
        {"enableCounters", js_enable_counters}, {"resetCounters", js_reset_counters},
        {"counters", js_counters}, {"accelStats", js_accel_stats}, {"lastTraceMs", js_last_trace_ms},
        {"setOption", js_set_option}, {"abiVersion", js_abi_version},
        {"traceAsync", js_trace_async}, {"syncAsync", js_sync_async},
        {"readRgba8Async", js_read_rgba8_async}, {"readAccumAsync", js_read_accum_async},
        {"readLatestRgba8", js_read_latest_rgba8}, {"latestSample", js_latest_sample}, {"readSampleRgba8", js_read_sample_rgba8},
        {"readSampleRgba8Async", js_read_sample_rgba8_async}, {"readSampleRgba8Into", js_read_sample_rgba8_into},
        {"pinHost", js_pin_host}, {"unpinHost", js_unpin_host},
        {"commUniqueId", js_comm_unique_id}, {"commInit", js_comm_init}, {"commPartition", js_comm_partition},
        {"commInfo", js_comm_info}, {"commDestroy", js_comm_destroy}, {"gather", js_gather},
        {"readFrameRgba8", js_read_frame_rgba8}, {"readFrameAccum", js_read_frame_accum}, {"imageSize", js_image_size},
        {"setRowBands", js_set_row_bands}, {"deviceBuffers", js_device_buffers}, {"frameDeviceBuffers", js_frame_device_buffers},
        {"bindOutput", js_bind_output}, {"setStream", js_set_stream},
        {"gatherAsync", js_gather_async}, {"readFrameRgba8Async", js_read_frame_rgba8_async}, {"readFrameAccumAsync", js_read_frame_accum_async},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
            napi_throw_error(env, NULL, "crt_napi: export failed");
            return NULL;
        }
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
