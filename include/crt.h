/*
 * crt.h -- C ABI of libcrt.so, the MI355X (gfx950) drop-in for the reference's
 * WebGPU path-trace compute pass.
 *
 * What it replaces (all paths under Meryx/ComputeRayTracer):
 *   - the device side:  src/shaders/ComputeShader.wgsl:77-117 (`main`) and
 *     src/shaders/UpdateVariables.wgsl:1-7 (`sample++`);
 *   - the host calls that feed and drive it in src/main.js:
 *       createBuffer/getMappedRange/unmap of bind-group-0 entries b4..b8
 *                                   (main.js:147-155,249-253,263-296,313-393)
 *       the zeroed accumulator + sample counter           (main.js:298-311)
 *       per-frame  dispatchWorkgroups(1) ; dispatchWorkgroups(ceil(W/8),ceil(H/8))
 *                                                          (main.js:598-611)
 *       uncapturederror reporting                          (main.js:11-14)
 *
 * Conventions: plain C, caller-owned host pointers, sizes in records unless
 * stated; every function returns 0 on success or a negative CRT_E* code and
 * leaves a message for crt_last_error().  One thread per context.  Work is
 * enqueued on the context's HIP stream; crt_sync / crt_read_* wait for it.
 * There is NO CPU fallback: without a HIP device crt_create fails.
 */
#ifndef CRT_H
#define CRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_ABI_VERSION 2

enum {
    CRT_OK = 0,
    CRT_EINVAL = -1,   /* bad argument / malformed buffer            */
    CRT_EDEVICE = -2,  /* HIP runtime error (message has the detail) */
    CRT_ESTATE = -3,   /* call out of order (e.g. trace before upload) */
    CRT_ENOMEM = -4
};

/* Acceleration modes for crt_build_accel. */
enum {
    CRT_ACCEL_NONE = 0, /* the reference's own loop over every primitive
                           (ComputeShader.wgsl:503-518), on the GPU        */
    CRT_ACCEL_BVH2 = 1, /* binned-SAH BVH2 built on the host; returns exactly
                           what the loop returns (closest t; equal t -> later
                           primitive) */
    CRT_ACCEL_LBVH = 2  /* the same structure built on the GPU, end to end (Morton order,
                           Karras hierarchy, collapse to 4-wide, quantisation, leaf-ordered
                           records): milliseconds instead of seconds to build (10 M
                           triangles: 0.11 s), same image, more nodes visited per ray  */
};

/* Counter slots for crt_counters(). */
enum {
    CRT_CNT_RAYS = 0,        /* intersect() invocations (primary+bounce+shadow) */
    CRT_CNT_NODES = 1,       /* BVH child boxes tested (2 per inner-node visit) */
    CRT_CNT_PRIMS = 2,       /* primitive intersection tests                    */
    CRT_CNT_PATHS = 3,       /* pixel-samples                                   */
    CRT_CNT_BOUNCES = 4,     /* path-loop iterations                            */
    CRT_CNT_SHADOW = 5,      /* shadow rays (subset of RAYS)                    */
    CRT_CNT_HITS = 6,        /* closest-hit attribute fetches                   */
    CRT_CNT_WALKED = 7,      /* rays that actually walked the BVH: RAYS minus shadow rays decided
                                without a walk (light's own primitive missed, or cos_theta == 0 so the
                                NEE term is exactly zero) -- wavefront pipeline only                  */
    CRT_NCOUNTERS = 8
};

typedef struct crt_ctx crt_ctx;

/* ~ navigator.gpu.requestAdapter()/requestDevice(), main.js:8-9. */
int crt_create(crt_ctx **out, int device_ordinal);
void crt_destroy(crt_ctx *ctx);
const char *crt_last_error(crt_ctx *ctx);   /* ctx may be NULL: last create error */
int crt_abi_version(void);

/* ~ the b4..b8 buffer uploads (main.js:147-393).  Byte layouts are exactly the
 * reference's: 80-byte Primitive records (category@0, data1@16, data2@32,
 * data3@48, data4@64 = emission_idx, reflectance_idx, material, index), camera
 * = 16 floats (eye,_,lookat,_,up,width,height,focal,_,_), spectra = nspectra
 * rows of 301 floats (last row = glass extinction), cie = 3 x 471 floats.
 * Requirements checked here: data4.w == array position (main.js:124,133
 * guarantees it), category in {0 patch,1 sphere,2 triangle}, material in
 * {0,1,2}, nlight >= 1.  Resets the accumulator and the accel structure. */
int crt_upload_scene(crt_ctx *ctx,
                     const void *primitives, size_t nprim,
                     const void *lights, size_t nlight,
                     const float *spectra, size_t nspectra,
                     const float *cie,
                     const float camera[16]);

/* Restrict this context to the pixel rectangle [x0,x1) x [y0,y1) of the full
 * W x H image (image-tile partition across GPUs).  Default: the whole image.
 * RNG seeds use the GLOBAL pixel coordinates (ComputeShader.wgsl:98), so a
 * tile is bit-identical to the same pixels of a full-frame render.  Resets the
 * accumulator. */
int crt_set_tile(crt_ctx *ctx, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);

/* Row-interleaved partition for load balance across GPUs: this context renders the rows y of
 * the full frame with (y / band_rows) % parts == part, full width, packed densely in its
 * buffers (local row j is global row ((j / band) * parts + part) * band + j % band).
 * Resets the accumulator.  crt_set_tile returns to a plain rectangle. */
int crt_set_row_bands(crt_ctx *ctx, uint32_t band_rows, uint32_t parts, uint32_t part);

int crt_build_accel(crt_ctx *ctx, int mode);

/* ~ main.js:298-311: zero alt_color_buffer, sample = 0. */
int crt_reset(crt_ctx *ctx);

/* ~ n_samples iterations of frame() (main.js:597-611):
 *   { sample++ (UpdateVariables.wgsl) ; path-trace dispatch (ComputeShader.wgsl) }.
 * Samples are accumulated into alt_color_buffer in order; the rgba8
 * framebuffer holds the tone-mapped average after the last one.
 * Asynchronous, and (option "wf_defer", default 1) PIPELINED across calls: a call publishes
 * its samples as a batch, enqueues the work the batch needs and returns without waiting for it;
 * up to "wf_ring" (default 32) batches are in flight, resolved in order under the following
 * crt_trace calls or in crt_sync.  A call only blocks for back-pressure (the ring is full).
 * Small calls are MERGED: their samples become one batch once "wf_cohort" (default 16) samples
 * have come together, or at crt_sync / any call that reads state (a batch of many samples per
 * pixel keeps the paths in flight inside a band of the image and runs 1.4x faster per sample;
 * the frames are the same bit for bit; "wf_cohort" = 1 makes every call its own batch).  So:
 *   - after crt_sync (or any crt_read_*, crt_counters, crt_last_*_ms) the buffers
 *     hold every sample requested so far;
 *   - in between, buffers bound with crt_bind_output hold, in stream order, the
 *     complete frame of an EARLIER crt_trace call -- at most wf_ring batches before the
 *     last one -- never a half-resolved one.  A display/gather loop that shows the latest
 *     complete frame while the next ones render needs no sync at all. */
int crt_trace(crt_ctx *ctx, uint32_t n_samples);
/* Finish everything requested so far and wait for it. */
int crt_sync(crt_ctx *ctx);

/* Current value of the `sample` counter (ComputeShader.wgsl:3). */
int crt_sample_count(crt_ctx *ctx, uint32_t *out);

/* Tile geometry: out[4] = x0, y0, width, height. */
int crt_tile(crt_ctx *ctx, uint32_t out[4]);

/* Readback of this context's tile, row-major, row 0 = top (the reference never
 * reads back; its blit pass is display-only).  accum: tw*th*4 floats (x,y,z,
 * pad -- the 16-byte stride of array<vec3<f32>>); rgba8: tw*th*4 bytes. */
int crt_read_accum(crt_ctx *ctx, float *out);
int crt_read_rgba8(crt_ctx *ctx, uint8_t *out);
/* The display step of the reference's frame loop (src/main.js:597-620 shows EVERY sample's frame: compute pass, blit,
 * requestAnimationFrame) without stopping the pipeline:
 *   crt_read_latest_rgba8   the newest COMPLETE frame in stream order and its sample index -- no flush: batches in
 *                           flight stay in flight (crt_read_rgba8 finishes everything first);
 *   crt_latest_sample       that index alone (non-blocking; retires what has finished meanwhile);
 *   crt_read_sample_rgba8   the frame as it was after exactly `sample` samples, from a ring of the last F frames (option
 *                           "frame_ring" = F, set before tracing; k_wf_resolve then stores every sample's frame, bit-identical
 *                           to a synced crt_trace(1) loop).  Waits only for the batch that holds the sample; a display loop
 *                           that requests frames a cohort ahead of the one it shows (host/display_loop.js) sees every frame
 *                           index exactly once while small calls are still merged into cohorts.  CRT_EINVAL for a sample that
 *                           has not been requested, has left the ring, or predates the context's last crt_reset /
 *                           crt_write_accum / change of "frame_ring" (a restored accumulator brings no frames with it). */
int crt_read_latest_rgba8(crt_ctx *ctx, uint8_t *out, uint32_t *sample);
/* Page-lock / release caller memory (hipHostRegister): readbacks into pinned memory run at PCIe speed -- for the frame
 * buffer a display loop reads every frame into.  No context needed; errors are reported through crt_last_error(NULL). */
int crt_pin_host(void *ptr, size_t bytes);
int crt_unpin_host(void *ptr);
int crt_latest_sample(crt_ctx *ctx, uint32_t *out);
int crt_read_sample_rgba8(crt_ctx *ctx, uint32_t sample, uint8_t *out);
/* Restore an accumulator + sample count (checkpoint/resume). */
int crt_write_accum(crt_ctx *ctx, const float *in, uint32_t sample);

/* Device pointers of the tile buffers (for an RCCL gather by the caller). */
int crt_device_buffers(crt_ctx *ctx, void **accum_dev, void **rgba8_dev);
/* Render into caller-owned DEVICE memory instead (e.g. a slice of a gather
 * buffer): accum_dev >= tw*th*16 B, rgba8_dev >= tw*th*4 B; NULL restores the
 * internal buffers. */
int crt_bind_output(crt_ctx *ctx, void *accum_dev, void *rgba8_dev);
/* Enqueue on the caller's hipStream_t instead of the context's own. */
int crt_set_stream(crt_ctx *ctx, void *hip_stream);

/* The context's device ordinal and the HIP stream it enqueues on (hipStream_t as void*), the full image's size
 * (out[2] = W, H): for code that composes with the context from outside, like the gather below. */
int crt_get_device(crt_ctx *ctx, int *out);
int crt_get_stream(crt_ctx *ctx, void **out);
int crt_image_size(crt_ctx *ctx, uint32_t out[2]);

/* ------------------------------------------------------------------ Multi-GPU: the frame across the ranks of a communicator
 * The reference drives ONE GPUDevice (src/main.js:8-9) and has no exchange step.  Here the frame is partitioned by rows
 * across `world` contexts -- one per GPU, as a rule one process per GPU -- the scene replicated, and the path's only
 * exchange is the gather of the finished strips: an RCCL all-gather over xGMI issued from libcrt on the context's stream
 * (SURVEY 8e).  Pixels depend on their global coordinates only (ComputeShader.wgsl:85-86,98), so the assembled frame is
 * bit-identical for every world size.
 *
 *   crt_comm_unique_id(id, CRT_COMM_RCCL)      on one rank; hand the 128 bytes to the others (IPC, a file, a socket)
 *   crt_comm_init(ctx, id, rank, world)        every rank, after crt_create (collective for RCCL)
 *   crt_upload_scene(ctx, ...)                 the same scene on every rank
 *   crt_comm_partition(ctx, band_rows)         this rank's rows: bands of band_rows rows dealt round-robin (8 balances
 *                                              regions of different path length), 0 = contiguous strips; the context then
 *                                              renders into strip buffers owned by the communicator (resets the accumulator)
 *   crt_build_accel; loop { crt_trace(ctx, n); crt_gather(ctx, CRT_GATHER_RGBA8); }     every rank
 *   crt_sync(ctx); crt_gather(ctx, CRT_GATHER_RGBA8 | CRT_GATHER_ACCUM);
 *   crt_read_frame_rgba8 / crt_read_frame_accum(ctx, out)       any rank: the whole W x H frame
 *
 * crt_gather is asynchronous and stream-ordered like crt_trace: it ships what the strip holds THEN -- after crt_sync
 * every sample requested so far, in a pipelined loop the latest complete frame (crt_trace's contract for bound outputs).
 * CRT_COMM_LOCAL is the same interface for contexts of ONE process (any devices): strips are exchanged with
 * device-to-device copies, no RCCL; every rank posts its crt_gather before any rank reads the frame. */
#define CRT_COMM_ID_BYTES 128
enum { CRT_COMM_RCCL = 0, CRT_COMM_LOCAL = 1 };
enum { CRT_GATHER_RGBA8 = 1, CRT_GATHER_ACCUM = 2 };
int crt_comm_unique_id(void *out_id /* CRT_COMM_ID_BYTES */, int transport);
int crt_comm_init(crt_ctx *ctx, const void *id, int rank, int world);
int crt_comm_partition(crt_ctx *ctx, uint32_t band_rows);
int crt_gather(crt_ctx *ctx, int what);
int crt_read_frame_rgba8(crt_ctx *ctx, uint8_t *out /* W*H*4, row 0 = top */);
int crt_read_frame_accum(crt_ctx *ctx, float *out /* W*H*4 floats */);
/* Device pointers of the assembled frame of the latest gather (valid until the next crt_comm_partition). */
int crt_frame_device_buffers(crt_ctx *ctx, void **accum_dev, void **rgba8_dev);
/* out[4] = rank, world, transport (-1: no communicator), rows of this rank. */
int crt_comm_info(crt_ctx *ctx, int out[4]);
int crt_comm_destroy(crt_ctx *ctx);          /* also done by crt_destroy */
/* The row layout itself (no GPU involved): the rows of the H-row frame that belong to `part` of `parts`, in local order;
 * n_rows receives their number, global_rows (may be NULL) their indices.  band_rows = 0: contiguous strips. */
int crt_layout_rows(uint32_t H, uint32_t band_rows, uint32_t parts, uint32_t part, uint32_t *n_rows, uint32_t *global_rows);

/* Counters accumulate over crt_trace calls while enabled (off by default: the
 * counting kernel variant is slower). */
int crt_enable_counters(crt_ctx *ctx, int on);
int crt_counters(crt_ctx *ctx, uint64_t out[CRT_NCOUNTERS]);
int crt_reset_counters(crt_ctx *ctx);

/* Device time from the start of the LAST crt_trace call to the end of its work
 * (HIP events on the context's stream; finishes the call's parked paths first),
 * and how many kernel launches that was. */
int crt_last_trace_ms(crt_ctx *ctx, float *ms, uint32_t *launches);

/* Device time of the DOMINANT kernel's launches, summed, and their number: the BVH traversal
 * kernel k_wf_trace of the wavefront pipeline -- every launch since the previous query (or
 * since option "time_kernels" was set: 1 brackets each launch with HIP events on the
 * stream it runs on, N > 1 also creates the event pairs for N launches up front) -- or the single trace kernel of the last crt_trace call in the
 * "pipeline"=0 form.  Syncs. */
int crt_last_kernel_ms(crt_ctx *ctx, float *ms, uint32_t *launches);

/* Tuning knobs.  "spp_per_launch": samples fused per batch (0 = default);
 * "pipeline": 1 = wavefront (default), 0 = single megakernel; "wf_pool": path slots
 * (0 = auto: a quarter of a batch, at least "wf_pool_spp" (8) slots per tile pixel, 1 M..24 M); "wf_waves_per_cu": persistent traversal
 * waves per CU and pipe; "wf_pipes": sub-pools on separate streams (1..4); "wf_defer": 0 = every crt_trace
 * call runs its paths to the end; "wf_ring" (2..32 batches in flight), "wf_cohort" (samples per batch that small calls are merged up to), "wf_chunk" (iterations enqueued at
 * a time), "wf_ahead" (iterations in flight per pipe before the call waits), "wf_feed_pct", "wf_finish_at",
 * "wf_flush_at", "wf_side_ppw", "wf_flush_ppw", "wf_tail_walk": pipeline tuning;
 * "quantize", "wf_width" (4 | 8: node width of the wavefront traversal; at crt_build_accel);
 * "wf_trace_form" (2: ray ring + primitive tasks, default; 1: the first traversal kernel); "frame_ring" = F (keep the
 * rgba8 frame of each of the last F samples for crt_read_sample_rgba8; 0 = off);
 * "time_kernels"; "debug_fail_alloc" = k (test hook: the k-th device allocation from now on reports
 * out of memory).  Setting an option first finishes what is in flight. */
int crt_set_option(crt_ctx *ctx, const char *name, int64_t value);

/* Accel statistics: out[0]=BVH2 inner nodes, [1]=leaves, [2]=max depth, [3]=device bytes,
 * [4]=bytes of node data fetched per child box tested by crt_trace (32: plain boxes, 16:
 * 16-bit quantised), [5]=node width crt_trace walks (2, 4 or 8), [6]=inner nodes of that tree,
 * [7]=builder (0: host binned SAH, 1: GPU LBVH). */
int crt_accel_stats(crt_ctx *ctx, uint64_t out[8]);

/* Test hooks: one closest-hit query per ray through the product's traversal
 * (rays: n x 8 floats ox,oy,oz,dx,dy,dz,exclude_as_u32_bits,_;  out: n x 8:
 * t, px,py,pz, nx,ny,nz, index_bits (0xFFFFFFFF = miss)); and elementwise
 * evaluation of the device math (fn codes: 0 sin 1 cos 2 exp 3 log2 4 exp2
 * 5 pow 6 sqrt 7 div 8 tan). */
int crt_debug_intersect(crt_ctx *ctx, const float *rays, size_t n, float *out);
int crt_debug_math(crt_ctx *ctx, int fn, const float *a, const float *b, float *out, size_t n);
/* Traversal-efficiency probes of the counting kernel variant (wave-level): inner iterations,
 * lanes active in them, leaf passes, lanes active in them, leaf loop trips, -, refills, lanes refilled. */
int crt_debug_probes(crt_ctx *ctx, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* CRT_H */
