// crt_comm.cpp -- the multi-GPU part of the C ABI (include/crt.h, "Multi-GPU"): the frame partitioned by rows across the
// ranks of a communicator, one context per GPU, and the path's ONE exchange step -- the gather of the finished strips --
// as an RCCL all-gather over xGMI issued from libcrt itself, so that any host above the C ABI (the Node addon as much as
// the Python one) reaches the multi-GPU configurations.  The reference has a single GPUDevice (src/main.js:8-9) and no
// exchange of any kind; SURVEY 8(e) defines this one.
//
// Written against the public C ABI only (crt_set_row_bands / crt_set_tile, crt_bind_output, crt_get_stream): a context
// does not know that it is part of a communicator, it renders its rows into the strip buffers bound here.
//
// Transports behind one interface:
//   RCCL   one process per GPU (or one thread per GPU): ncclCommInitRank + ncclAllGather.  librccl is loaded on first
//          use (dlopen), so single-GPU hosts never pay for it.
//   local  contexts of ONE process (any devices, the same one included): every rank copies its strip into every rank's
//          gather buffer with hipMemcpyPeerAsync and the ranks' streams are joined by events.  What one process driving
//          several GPUs uses, and what lets the partition / gather / assembly path be tested on a one-GPU box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/crt.h"

namespace crt {
hipError_t launch_assemble(const void *full, void *frame, uint32_t elem_bytes, uint32_t W, uint32_t H, uint32_t world, uint32_t rows_max,
                           uint32_t band, hipStream_t stream);
}
extern "C" int crt_internal_fail(crt_ctx *c, int code, const char *msg);

namespace {

// ---------------------------------------------------------------- RCCL, loaded on demand
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("librccl could not be loaded: ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString) {
            err = "librccl lacks an expected entry point";
            dlclose(lib); lib = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_mu;                                   // guards g_rccl, g_comms and g_groups (never held across a GPU wait)

// ---------------------------------------------------------------- the local transport's shared state
struct Comm;
struct LocalGroup {
    int world = 0;
    std::vector<Comm *> member;                    // by rank; null until that rank has joined
    std::vector<uint64_t> posted_rgba, posted_accum;   // per rank: gathers posted so far
};

struct Comm {
    crt_ctx *ctx = nullptr;
    int rank = 0, world = 1, device = 0;
    bool local = false;
    ncclComm_t nccl = nullptr;
    std::shared_ptr<LocalGroup> group;
    char id[CRT_COMM_ID_BYTES] = {0};
    // partition and buffers (valid once crt_comm_partition has run)
    bool partitioned = false;
    uint32_t W = 0, H = 0, band = 0, rows = 0, rows_max = 0;
    void *strip_accum = nullptr, *strip_rgba = nullptr;      // this rank's rows, padded to rows_max (bound as the context's output)
    void *full_accum = nullptr, *full_rgba = nullptr;        // [world][rows_max][W]: the gathered strips
    void *frame_accum = nullptr, *frame_rgba = nullptr;      // [H][W]: assembled frame
    uint64_t n_rgba = 0, n_accum = 0;                        // gathers posted by this rank
    uint64_t asm_rgba = 0, asm_accum = 0;                    // ... and assembled
    hipEvent_t ev_rgba = nullptr, ev_accum = nullptr;        // local transport: after this rank's copies of its latest gather
    hipEvent_t ev_asm_rgba = nullptr, ev_asm_accum = nullptr;  // ... and after this rank's latest assembly (it reads `full`: a peer's next copy waits for it)
    bool asm_rgba_recorded = false, asm_accum_recorded = false;
};
std::map<crt_ctx *, std::unique_ptr<Comm>> g_comms;
std::map<std::string, std::weak_ptr<LocalGroup>> g_groups;
std::atomic<uint64_t> g_local_seq{1};

int fail(crt_ctx *c, int code, const std::string &msg) { return crt_internal_fail(c, code, msg.c_str()); }
#define CHIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, e_ == hipErrorOutOfMemory ? CRT_ENOMEM : CRT_EDEVICE, std::string(#call ": ") + hipGetErrorString(e_)); } while (0)

Comm *find(crt_ctx *c)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_comms.find(c);
    return it == g_comms.end() ? nullptr : it->second.get();
}

void free_buffers(Comm &m)
{
    for (void **p : {&m.strip_accum, &m.strip_rgba, &m.full_accum, &m.full_rgba, &m.frame_accum, &m.frame_rgba})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    m.partitioned = false;
}

bool is_local_id(const void *id) { return std::memcmp(id, "CRTLOCAL", 8) == 0; }

// rows of the frame that belong to `part` (the layout crt_set_row_bands / contiguous strips give)
uint32_t rows_of(uint32_t H, uint32_t band, uint32_t parts, uint32_t part)
{
    uint32_t n = 0;
    (void)crt_layout_rows(H, band, parts, part, &n, nullptr);
    return n;
}

}  // namespace

extern "C" {

int crt_layout_rows(uint32_t H, uint32_t band_rows, uint32_t parts, uint32_t part, uint32_t *n_rows, uint32_t *global_rows)
{
    if (parts == 0 || part >= parts || !n_rows) return CRT_EINVAL;
    uint32_t n = 0;
    if (band_rows == 0) {                                      // contiguous strips of ceil(H / parts) rows
        const uint32_t per = (H + parts - 1) / parts;
        const uint32_t y0 = std::min<uint64_t>((uint64_t)part * per, H), y1 = std::min<uint64_t>((uint64_t)y0 + per, H);
        for (uint32_t y = y0; y < y1; y++, n++) if (global_rows) global_rows[n] = y;
    } else {                                                   // bands of band_rows rows dealt round-robin
        for (uint64_t b = part; b * band_rows < H; b += parts)
            for (uint32_t y = (uint32_t)(b * band_rows); y < H && y < (b + 1) * band_rows; y++, n++)
                if (global_rows) global_rows[n] = y;
    }
    *n_rows = n;
    return CRT_OK;
}

int crt_comm_unique_id(void *out, int transport)
{
    if (!out) return CRT_EINVAL;
    std::memset(out, 0, CRT_COMM_ID_BYTES);
    if (transport == CRT_COMM_LOCAL) {
        const uint64_t seq = g_local_seq.fetch_add(1);
        std::memcpy(out, "CRTLOCAL", 8);
        std::memcpy((char *)out + 8, &seq, sizeof seq);
        return CRT_OK;
    }
    if (transport != CRT_COMM_RCCL) return crt_internal_fail(nullptr, CRT_EINVAL, "crt_comm_unique_id: unknown transport");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_rccl.load()) return crt_internal_fail(nullptr, CRT_EDEVICE, ("crt_comm_unique_id: " + g_rccl.err).c_str());
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return crt_internal_fail(nullptr, CRT_EDEVICE, (std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r)).c_str());
    static_assert(sizeof(ncclUniqueId) == CRT_COMM_ID_BYTES, "id size");
    std::memcpy(out, &id, sizeof id);
    return CRT_OK;
}

int crt_comm_init(crt_ctx *c, const void *id, int rank, int world)
{
    if (!c || !id) return CRT_EINVAL;
    if (world < 1 || rank < 0 || rank >= world || world > 1024) return fail(c, CRT_EINVAL, "crt_comm_init: need 0 <= rank < world <= 1024");
    if (find(c)) return fail(c, CRT_ESTATE, "crt_comm_init: the context already has a communicator (crt_comm_destroy first)");
    int device = 0;
    { int rc = crt_get_device(c, &device); if (rc) return rc; }
    CHIP(c, hipSetDevice(device));
    std::unique_ptr<Comm> m(new Comm());
    m->ctx = c; m->rank = rank; m->world = world; m->device = device;
    std::memcpy(m->id, id, CRT_COMM_ID_BYTES);
    m->local = is_local_id(id);
    // (the events first: nothing below may fail once the rank has joined its group or its RCCL communicator exists)
    auto drop_events = [&]() {
        for (hipEvent_t *e : {&m->ev_rgba, &m->ev_accum, &m->ev_asm_rgba, &m->ev_asm_accum}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    };
    if (hipEventCreateWithFlags(&m->ev_rgba, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&m->ev_accum, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_asm_rgba, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&m->ev_asm_accum, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        drop_events();
        return fail(c, CRT_EDEVICE, "crt_comm_init: hipEventCreate failed");
    }
    if (m->local) {
        std::lock_guard<std::mutex> lk(g_mu);
        const std::string key((const char *)id, CRT_COMM_ID_BYTES);
        std::shared_ptr<LocalGroup> g = g_groups[key].lock();
        if (!g) {
            g = std::make_shared<LocalGroup>();
            g->world = world; g->member.assign((size_t)world, nullptr);
            g->posted_rgba.assign((size_t)world, 0); g->posted_accum.assign((size_t)world, 0);
            g_groups[key] = g;
        }
        if (g->world != world) { drop_events(); return fail(c, CRT_EINVAL, "crt_comm_init: the ranks of one id disagree about the world size"); }
        if (g->member[(size_t)rank]) { drop_events(); return fail(c, CRT_EINVAL, "crt_comm_init: that rank of the id is taken"); }
        g->member[(size_t)rank] = m.get();
        m->group = g;
    } else {
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if (!g_rccl.load()) { drop_events(); return fail(c, CRT_EDEVICE, "crt_comm_init: " + g_rccl.err); }
        }
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        const ncclResult_t r = g_rccl.CommInitRank(&m->nccl, world, uid, rank);   // (collective: every rank of the id calls it)
        if (r != ncclSuccess) { drop_events(); return fail(c, CRT_EDEVICE, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_comms[c] = std::move(m);
    return CRT_OK;
}

int crt_comm_destroy(crt_ctx *c)
{
    std::unique_ptr<Comm> m;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_comms.find(c);
        if (it == g_comms.end()) return CRT_OK;
        m = std::move(it->second);
        g_comms.erase(it);
        if (m->group) m->group->member[(size_t)m->rank] = nullptr;
    }
    (void)hipSetDevice(m->device);
    void *st = nullptr;
    if (crt_get_stream(c, &st) == CRT_OK) (void)hipStreamSynchronize((hipStream_t)st);
    if (m->partitioned) (void)crt_bind_output(c, nullptr, nullptr);
    free_buffers(*m);
    if (m->nccl) (void)g_rccl.CommDestroy(m->nccl);
    if (m->ev_rgba) (void)hipEventDestroy(m->ev_rgba);
    if (m->ev_accum) (void)hipEventDestroy(m->ev_accum);
    if (m->ev_asm_rgba) (void)hipEventDestroy(m->ev_asm_rgba);
    if (m->ev_asm_accum) (void)hipEventDestroy(m->ev_asm_accum);
    return CRT_OK;
}

// called by crt_destroy
void crt_comm_on_destroy(crt_ctx *c) { (void)crt_comm_destroy(c); }

int crt_comm_partition(crt_ctx *c, uint32_t band_rows)
{
    Comm *m = find(c);
    if (!m) return fail(c, CRT_ESTATE, "crt_comm_partition: crt_comm_init first");
    uint32_t wh[2];
    { int rc = crt_image_size(c, wh); if (rc) return rc; }
    const uint32_t W = wh[0], H = wh[1];
    if (band_rows > 65536u) return fail(c, CRT_EINVAL, "crt_comm_partition: band_rows must be 0 (contiguous strips) .. 65536");
    CHIP(c, hipSetDevice(m->device));
    if (m->partitioned) { int rc = crt_bind_output(c, nullptr, nullptr); if (rc) return rc; }
    free_buffers(*m);
    uint32_t rows_max = 0;
    for (int p = 0; p < m->world; p++) rows_max = std::max(rows_max, rows_of(H, band_rows, (uint32_t)m->world, (uint32_t)p));
    m->W = W; m->H = H; m->band = band_rows; m->rows_max = rows_max;
    m->rows = rows_of(H, band_rows, (uint32_t)m->world, (uint32_t)m->rank);
    int rc;
    if (band_rows) rc = crt_set_row_bands(c, band_rows, (uint32_t)m->world, (uint32_t)m->rank);
    else {
        const uint32_t per = (H + (uint32_t)m->world - 1) / (uint32_t)m->world;
        const uint32_t y0 = std::min<uint64_t>((uint64_t)m->rank * per, H);
        rc = crt_set_tile(c, 0, y0, W, std::min<uint64_t>((uint64_t)y0 + per, H));
    }
    if (rc) return rc;
    const size_t strip_px = (size_t)std::max(rows_max, 1u) * W, frame_px = (size_t)std::max(H, 1u) * W;
    CHIP(c, hipMalloc(&m->strip_accum, strip_px * 16)); CHIP(c, hipMalloc(&m->strip_rgba, strip_px * 4));
    CHIP(c, hipMalloc(&m->full_accum, strip_px * 16 * (size_t)m->world)); CHIP(c, hipMalloc(&m->full_rgba, strip_px * 4 * (size_t)m->world));
    CHIP(c, hipMalloc(&m->frame_accum, frame_px * 16)); CHIP(c, hipMalloc(&m->frame_rgba, frame_px * 4));
    CHIP(c, hipMemset(m->strip_accum, 0, strip_px * 16)); CHIP(c, hipMemset(m->strip_rgba, 0, strip_px * 4));
    CHIP(c, hipMemset(m->frame_accum, 0, frame_px * 16)); CHIP(c, hipMemset(m->frame_rgba, 0, frame_px * 4));
    m->partitioned = true;
    m->n_rgba = m->n_accum = m->asm_rgba = m->asm_accum = 0;
    if (m->group) { std::lock_guard<std::mutex> lk(g_mu); m->group->posted_rgba[(size_t)m->rank] = 0; m->group->posted_accum[(size_t)m->rank] = 0; }
    // the context renders straight into the padded strips (and its accumulator starts at zero there)
    return crt_bind_output(c, m->strip_accum, m->strip_rgba);
}

// Post one gather of this rank's strip on the context's stream: what is in the strip THEN, in stream order -- after a
// crt_sync every sample requested so far, in the middle of a pipelined run the latest complete frame (crt_trace's contract
// for bound outputs).
static int gather_one(crt_ctx *c, Comm *m, bool accum)
{
    void *st = nullptr;
    { int rc = crt_get_stream(c, &st); if (rc) return rc; }
    hipStream_t s = (hipStream_t)st;
    const size_t elem = accum ? 16 : 4, strip_bytes = (size_t)std::max(m->rows_max, 1u) * m->W * elem;
    const void *src = accum ? m->strip_accum : m->strip_rgba;
    void *dst = accum ? m->full_accum : m->full_rgba;
    if (!m->local) {
        const ncclResult_t r = g_rccl.AllGather(src, dst, strip_bytes, ncclUint8, m->nccl, s);
        if (r != ncclSuccess) return fail(c, CRT_EDEVICE, std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
        (accum ? m->n_accum : m->n_rgba)++;
        return CRT_OK;
    }
    // local transport: this rank's strip into slot `rank` of every member's gather buffer
    std::vector<Comm *> peers;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        peers = m->group->member;
    }
    for (Comm *p : peers) {
        if (!p || !p->partitioned) return fail(c, CRT_ESTATE, "crt_gather: every rank of the communicator must have called crt_comm_partition");
        if (p->rows_max != m->rows_max || p->W != m->W || p->H != m->H || p->band != m->band)
            return fail(c, CRT_EINVAL, "crt_gather: the ranks of the communicator disagree about the partition");
        char *pd = (char *)(accum ? p->full_accum : p->full_rgba) + (size_t)m->rank * strip_bytes;
        // (the peer's assembly of the PREVIOUS gather reads the buffer this copy writes: behind it, whatever the streams)
        if (p != m && (accum ? p->asm_accum_recorded : p->asm_rgba_recorded)) CHIP(c, hipStreamWaitEvent(s, accum ? p->ev_asm_accum : p->ev_asm_rgba, 0));
        CHIP(c, hipMemcpyPeerAsync(pd, p->device, src, m->device, strip_bytes, s));
    }
    CHIP(c, hipEventRecord(accum ? m->ev_accum : m->ev_rgba, s));
    std::lock_guard<std::mutex> lk(g_mu);
    (accum ? m->n_accum : m->n_rgba)++;
    (accum ? m->group->posted_accum : m->group->posted_rgba)[(size_t)m->rank] = accum ? m->n_accum : m->n_rgba;
    return CRT_OK;
}

int crt_gather(crt_ctx *c, int what)
{
    Comm *m = find(c);
    if (!m || !m->partitioned) return fail(c, CRT_ESTATE, "crt_gather: crt_comm_init and crt_comm_partition first");
    if (!(what & (CRT_GATHER_RGBA8 | CRT_GATHER_ACCUM))) return fail(c, CRT_EINVAL, "crt_gather: nothing to gather");
    CHIP(c, hipSetDevice(m->device));
    if (what & CRT_GATHER_RGBA8) { int rc = gather_one(c, m, false); if (rc) return rc; }
    if (what & CRT_GATHER_ACCUM) { int rc = gather_one(c, m, true); if (rc) return rc; }
    return CRT_OK;
}

// Assemble the frame of the latest gather on this rank's stream (full -> frame: the rows back in image order).
static int assemble(crt_ctx *c, Comm *m, bool accum, hipStream_t s)
{
    uint64_t &done = accum ? m->asm_accum : m->asm_rgba;
    const uint64_t posted = accum ? m->n_accum : m->n_rgba;
    if (posted == 0) return fail(c, CRT_ESTATE, "crt_read_frame: no crt_gather of that buffer yet");
    if (m->local) {
        // every rank's copies of gather number `posted` must be in: their streams are joined through their events
        std::vector<Comm *> peers;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            for (int p = 0; p < m->world; p++)
                if ((accum ? m->group->posted_accum : m->group->posted_rgba)[(size_t)p] < posted)
                    return fail(c, CRT_ESTATE, "crt_read_frame: rank " + std::to_string(p) + " of the communicator has not posted its crt_gather yet");
            peers = m->group->member;
        }
        for (Comm *p : peers) if (p && p != m) CHIP(c, hipStreamWaitEvent(s, accum ? p->ev_accum : p->ev_rgba, 0));
    }
    if (done != posted) {
        const hipError_t e = crt::launch_assemble(accum ? m->full_accum : m->full_rgba, accum ? m->frame_accum : m->frame_rgba, accum ? 16u : 4u,
                                                  m->W, m->H, (uint32_t)m->world, std::max(m->rows_max, 1u), m->band, s);
        if (e != hipSuccess) return fail(c, CRT_EDEVICE, std::string("frame assembly: ") + hipGetErrorString(e));
        done = posted;
        if (m->local) {
            CHIP(c, hipEventRecord(accum ? m->ev_asm_accum : m->ev_asm_rgba, s));
            (accum ? m->asm_accum_recorded : m->asm_rgba_recorded) = true;
        }
    }
    return CRT_OK;
}

static int read_frame(crt_ctx *c, void *out, bool accum)
{
    Comm *m = find(c);
    if (!m || !m->partitioned) return fail(c, CRT_ESTATE, "crt_read_frame: crt_comm_init and crt_comm_partition first");
    if (!out) return CRT_EINVAL;
    CHIP(c, hipSetDevice(m->device));
    void *st = nullptr;
    { int rc = crt_get_stream(c, &st); if (rc) return rc; }
    hipStream_t s = (hipStream_t)st;
    { int rc = assemble(c, m, accum, s); if (rc) return rc; }
    const size_t bytes = (size_t)m->W * m->H * (accum ? 16 : 4);
    if (bytes) CHIP(c, hipMemcpyAsync(out, accum ? m->frame_accum : m->frame_rgba, bytes, hipMemcpyDeviceToHost, s));
    CHIP(c, hipStreamSynchronize(s));
    return CRT_OK;
}

int crt_read_frame_rgba8(crt_ctx *c, uint8_t *out) { return read_frame(c, out, false); }
int crt_read_frame_accum(crt_ctx *c, float *out) { return read_frame(c, out, true); }

int crt_frame_device_buffers(crt_ctx *c, void **accum_dev, void **rgba8_dev)
{
    Comm *m = find(c);
    if (!m || !m->partitioned) return fail(c, CRT_ESTATE, "crt_frame_device_buffers: crt_comm_init and crt_comm_partition first");
    CHIP(c, hipSetDevice(m->device));
    void *st = nullptr;
    { int rc = crt_get_stream(c, &st); if (rc) return rc; }
    if (accum_dev) { if (m->n_accum) { int rc = assemble(c, m, true, (hipStream_t)st); if (rc) return rc; } *accum_dev = m->frame_accum; }
    if (rgba8_dev) { if (m->n_rgba) { int rc = assemble(c, m, false, (hipStream_t)st); if (rc) return rc; } *rgba8_dev = m->frame_rgba; }
    return CRT_OK;
}

int crt_comm_info(crt_ctx *c, int out[4])
{
    if (!out) return CRT_EINVAL;
    Comm *m = find(c);
    if (!m) { out[0] = 0; out[1] = 1; out[2] = -1; out[3] = 0; return CRT_OK; }
    out[0] = m->rank; out[1] = m->world; out[2] = m->local ? CRT_COMM_LOCAL : CRT_COMM_RCCL; out[3] = (int)m->rows;
    return CRT_OK;
}

}  // extern "C"
