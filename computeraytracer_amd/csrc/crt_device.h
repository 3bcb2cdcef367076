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
constexpr int kStackDepth = 64;     // single-ray BVH2 walk: the SAH builder stops at depth 30, a GPU LBVH can reach 62
constexpr uint32_t kNLambda = 301;
constexpr uint32_t kNCie = 471;
constexpr int CRT_NCOUNTERS_DEV = 16;   // 8 public (crt_counters) + 8 traversal-efficiency probes

struct DevScene {
    const float4 *prim;
    const float4 *primD;
    const uint32_t *slot_of_index;
    const float4 *nodes;
    const float4 *nodes4;       // 8 x float4 per 4-wide node (128 B)
    const uint4 *nodes4q;       // 4 x uint4 per quantised 4-wide node (64 B), or null
    const uint4 *nodes8q;       // 8 x uint4 per quantised 8-wide node (128 B = one L2 line), or null
    const float *spectra;
    const float *cie;
    const float4 *lights;
    uint32_t nprim;
    uint32_t npatch;       // planar patches (category 0) in the scene: with none, nobody needs primD
    int32_t root;
    int32_t root4;
    int32_t root8;
    uint32_t n_nodes4;
    uint32_t nspectra;
    uint32_t nlight;
    float hit_pad;
    uint32_t nf_last[2];   // array positions of the last and the second-to-last patch / sphere (kNoHit: none): what the
                           // reference's loop returns for a ray with a NaN in its direction (crt_wavefront.hip, NEE)
    float inv_nlight;      // 1.0f / f32(nlight)
    uint32_t W, H;         // full image
    float qbase[3], qscale[3];  // plane = qbase + q * qscale (quantised nodes)
    float cam[12];         // llc, horizontal, vertical, eye  (ComputeShader.wgsl:470-487 hoisted)
};

struct TraceParams {
    DevScene sc;
    uint32_t x0, y0, tw, th;        // tile rectangle (local buffer is tw x th)
    uint32_t band, stride, phase;   // row interleave: global y = y0 + (ly/band)*band*stride + phase*band + ly%band
    uint32_t first_sample, n_samples;
    float4 *accum;
    uchar4 *rgba;
    unsigned long long *counters;   // CRT_NCOUNTERS, may be null
    uint32_t tiles_x, tiles_y;
};


// ---------------------------------------------------------------------------------------------
// Wavefront pipeline state (crt_wavefront.hip).  A pool of P path slots lives in HBM (SoA,
// one float4/uint4 per slot per array, so every access is a coalesced 16-byte stream):
//   ray_o   (o.xyz, exclude bits)         next extension ray of the slot
//   ray_d   (d.xyz, -)
//   sh_d    (light dir.xyz, t of the light's own primitive)   pending shadow ray
//   beta, radiance, nee                   path throughput, radiance, NEE term waiting for visibility
//   rng     uvec4 PCG state               (ComputeShader.wgsl:899)
//   misc    (work id, flags, last_bounce_pdf bits, etaScale bits)
//   hit     (t, hit slot bits)            written by the trace kernel for extension rays
//   vis     in: light primitive index, out: 1 = light visible    (shadow rays)
//   list[parity][class]                   slots with an active ray this iteration (ballot/popc compacted)
//   staging[b] (xyz, -) per (sample, pixel)  finished samples of the batch with id b, summed in sample order by k_wf_resolve
constexpr uint32_t kWfAlive = 1u, kWfDying = 2u, kWfShadow = 4u, kWfSpecular = 8u, kWfInTrans = 16u;
constexpr uint32_t kWfNanRay = 1u << 31;   // the slot's (camera) ray is non-finite and has not been traced: the next shade step resolves it with the reference loop
constexpr uint32_t kWfHasRad = 1u << 30;   // radiance[slot] holds the path's radiance (else it is still zero: nothing was ever stored)
// ray-list entries: the slot, and on shadow-list entries a mark "this slot also listed an extension ray"
constexpr uint32_t kWfListSlot = 0x7FFFFFFFu, kWfListAlsoExt = 0x80000000u;
constexpr uint32_t kWfDepthShift = 8, kWfLambdaShift = 16;      // depth: 8 bits, lambda0: 9 bits
constexpr uint32_t kWfBatchShift = 25;                          // 5 bits: which of the (up to kWfRing) batches in flight the path belongs to
constexpr uint32_t kWfRing = 32;                                // batch ids cycle 0..ring-1 (ring <= kWfRing): queues, staging buffers, side pools are indexed by id
// Stragglers: once only a few paths of a batch are left (its queue is long empty, the pool is busy with the
// next batch), k_wf_shade moves them out of the pool into a small side pool, where k_wf_finish runs them to
// their end on another stream.  Side-pool slots precede the pool in the same arrays:
// [(id*pipes + pipe)*kWfSideCap, +kWfSideCap).
constexpr uint32_t kWfSideCap = 65536;
constexpr uint32_t kWfOverflowLevels = 96;                   // global stack levels per traversal lane beyond the LDS entries

// Every queue counter is sharded kWfShards ways, one 128-byte line per shard: same-address
// returning atomics serialize at ~11 ns each on gfx950, which at one atomic per wave would
// cost more than the kernels themselves.  Shade block b appends to shard b % kWfShards of
// the ray lists; traversal waves and re-arming waves pick a non-empty shard with one
// wave-wide load + ballot.
constexpr uint32_t kWfShards = 64;
// rays listed by shade per class / fetch cursor of trace / slots still alive after the shade launch, per batch id
// n_dead: slots this shard's shade blocks found or left dead in the launch (k_wf_gen re-arms them by whole waves)
struct WfShard { uint32_t n[4], cur, n_dead, pad0[26], alive[kWfRing]; };
static_assert(sizeof(WfShard) == 256, "two lines per shard: the list counters and cursor, the per-batch counts");
struct WfWork { uint32_t cur, pad[31]; };                    // next work item of this shard's range
struct WfCtl {                       // device control block, one per context
    WfShard shard[4][kWfShards];     // ring-indexed by iteration & 3 (it-1 is read, it written, it+1 zeroed)
    unsigned long long counters[kWfShards][CRT_NCOUNTERS_DEV];   // sharded like everything the waves add to (a shard = one 128-byte line); the host sums
    uint32_t side_count[kWfRing];    // paths moved to the side pool, per batch id
    uint32_t dropped;                // paths a capacity guard had to leave behind (side pool full, bounce guard of
                                     // k_wf_finish): must stay 0 -- the host turns anything else into CRT_EDEVICE
};
// The work queue is shared by the pipes of a context (two half-pools run on two streams so that
// one half's streaming shade pass overlaps the other half's latency-bound traversal).  There are kWfRing
// of them, one per batch id: the next batch's work is published while the current batch's queue
// still holds a few iterations' worth, so the pool never runs dry between batches.
struct WfWorkQ {
    WfWork work[kWfShards];
    uint32_t work_done;              // set once every work shard is exhausted (saves the scans)
    uint32_t work_per_shard;         // the queue's extent (k_wf_init copies WfSeg here for write_status)
    unsigned long long work_total;
    uint32_t pad_[28];
};
static_assert(sizeof(WfWorkQ) == (kWfShards + 1) * 128, "whole lines");
// One batch's share of the parameters (indexed by batch id like the queues and the staging buffers).
struct WfSeg {
    unsigned long long work_total;   // n_samples * npix_padded
    uint32_t work_per_shard;         // work items per shard (multiple of 64)
    uint32_t first_sample;
};

// What the host needs to know about an iteration, reduced on the device (write_status in crt_wavefront.hip) into one
// small record in pinned host memory.
struct WfStatus {
    uint32_t it_end;                       // the status describes the state after iterations < it_end of this pipe
    uint32_t dropped;                      // WfCtl::dropped
    uint32_t bound;                        // largest number of rays one shard listed in the last iteration
    uint32_t pad_;
    unsigned long long rays;               // rays listed in the last iteration (= alive slots once the queues are dry)
    unsigned long long consumed[kWfRing];  // work items taken from that batch's queue so far (all pipes)
    uint32_t left[kWfRing];                // 1: that batch's queue still holds work
    uint32_t alive[kWfRing];               // paths of that batch alive in THIS pipe's pool after its last shade launch
};

// k_wf_finish runs the side-pool paths of several (batch, pipe) pairs in one launch: segment s = blocks
// [s * blocks_per_seg, (s + 1) * blocks_per_seg), paths side_base .. + min(side_count, kWfSideCap) of that pipe's control block.
constexpr uint32_t kWfFinishSegs = 32;
struct WfFinishSegs {
    struct WfCtl *ctl[kWfFinishSegs];
    uint32_t base[kWfFinishSegs], batch[kWfFinishSegs];
    uint32_t n, blocks_per_seg;
};

struct WfParams {
    DevScene sc;
    float4 *ray_o, *ray_d, *sh_d, *beta, *radiance, *nee;
    uint4 *rng, *misc;
    float2 *hit;
    uint32_t *vis;
    // The rays of an iteration as compacted records, [iteration parity][class: camera, bounce, shadow of camera hit, shadow],
    // entry i of a shard's list at [(parity*4 + class) * list_cap*kWfShards + shard*list_cap + i]:
    //   recA (o.xyz, exclude bits)   recB extension ray: (d.xyz, slot | kWfListAlsoExt if already resolved)
    //                                     shadow ray:    (light dir.xyz, t of the light's own primitive)
    //   recC shadow rays only: (slot | kWfListAlsoExt if the slot also listed an extension ray, light primitive index, its slot, -)
    // (the records ARE the ray lists: tail mode walks recB.w / recC.x of the previous iteration)
    float4 *recA, *recB;
    uint4 *recC;
    uint32_t *dead;                  // [shard * list_cap + i]: slots that are dead after this iteration's shade launch (k_wf_gen's input)
    uint32_t rearm;                  // k_wf_shade: list the dead slots (a k_wf_gen launch follows: some queue may hold work)
    uint32_t gen_blocks;             // k_wf_gen: blocks per shard (block j of a shard takes chunks j, j + gen_blocks, ... of its dead list)
    float4 *staging[kWfRing];        // finished samples, per batch id (several batches can be in flight)
    uint32_t batch_id;           // id of the newest batch (k_wf_init: the batch being set up; k_wf_resolve / k_wf_finish: the batch to resolve / finish)
    WfStatus *status_out;            // k_wf_shade: where its first wave writes the PREVIOUS iteration's status record (pinned host memory), or null
    uint32_t count_alive;            // k_wf_shade: count the alive paths per batch id (WfShard::alive) -- needed once more than one batch is in flight
    uint32_t keep_pool;              // k_wf_init: leave the slots and list counters alone (paths of the batches before live on)
    uint32_t side_base[kWfRing];     // first side-pool slot of this pipe, per batch id
    uint32_t evict_mask;             // k_wf_shade: bit b = move the alive paths of batch id b to the side pool first
    WfCtl *ctl;
    WfWorkQ *wq;                     // [kWfRing], by batch id
    WfSeg seg[kWfRing];
    uint32_t seg_order[kWfRing];     // ids of the queues dead slots re-arm from, oldest first
    uint32_t seg_n;                  // how many of them
    uint32_t slot_base;              // this pipe's slots are [slot_base, slot_base + P)
    uint32_t reset_wq;               // k_wf_init also resets the work queue of batch_id
    uint32_t P;                      // slots of this pipe
    uint32_t x0, y0, tw, th;         // tile rectangle (local buffer is tw x th)
    uint32_t band, stride, phase;    // row interleave: global y = y0 + (ly/band)*band*stride + phase*band + ly%band
    uint32_t tiles_x, tiles_y;
    uint32_t npix_padded;            // tiles_x*tiles_y*64: work item = sample_off * npix_padded + tile*64 + lane
    uint32_t list_cap;               // list entries per shard
    uint32_t tail_bound;             // 0: one shade thread per slot; >0: tail mode, threads walk the previous
                                     //    iteration's ray lists (<= tail_bound entries per shard and list)
    uint32_t n_samples;              // k_wf_resolve: samples of the batch to resolve
    float4 *accum;
    uchar4 *rgba;
    uchar4 *frames;                  // k_wf_resolve: ring of `frame_ring` tile-sized rgba8 frames, frame of sample s at [(s - 1) % frame_ring], or null
    uint32_t frame_ring;
    const uint32_t *tea;             // per tile pixel: tea(px, py*100), the 16-round seed of the pixel's RNG (:98), computed once
    uint32_t count;                  // 1: maintain ctl->counters
    uint32_t trace_form;             // 2: k_wf_trace2 (ray ring + primitive tasks) where the tree is the quantised 4-wide one; else k_wf_trace
    int *stack_overflow;             // [level - LDS entries][global lane], for stacks deeper than the LDS part (kWfOverflowLevels levels)
    uint32_t overflow_lanes;
};

}  // namespace crt
