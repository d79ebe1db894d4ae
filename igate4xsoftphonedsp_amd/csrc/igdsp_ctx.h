// Private to the C-ABI translation units (igdsp_capi.hip, igdsp_io.hip): the context object and the small helpers every
// entry uses.  Not part of the ABI (include/igdsp.h is).
#pragma once
#include "igdsp_internal.h"

#include <atomic>
#include <cstdio>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
constexpr uint32_t kSlot = IGDSP_MAX_PAYLOAD;          // staging slot bytes (tp_adapter::payload_buff[256])
constexpr uint32_t kStageDepth = IGDSP_STAGE_DEPTH;    // frames per channel between two flushes (8 x 20 ms)
constexpr uint32_t kNoChan = 0xFFFFFFFFu;
constexpr int32_t kDirectCalls = 1 << 16;              // pjsua_call_id values are small non-negative ints
}  // namespace

struct igdsp_ctx {
    int device = -1;
    int cus = 0;
    std::string name;
    uint32_t max_channels = 0;
    hipStream_t stream = nullptr;
    int variant = 0;
    std::string err;

    // bulk output buffers handed out by igdsp_io_alloc with their halves in two memory classes: a kernel that writes one may
    // pick the traversal that suits that placement (launch_roundtrip)
    struct IoRange { const char *base; size_t bytes; };
    std::vector<IoRange> spread_ranges;
    // igdsp_io_alloc's memory of what it learnt: spare 128 MiB chunks of physical memory whose class a search (or a freed buffer
    // set) has established, kept mapped nowhere, up to io_spare_cap per class.  Class of a chunk = a property of the physical
    // memory, so only chunks that are HELD keep it; a later igdsp_io_alloc that these cover maps them without a single probe
    // (and releases nothing, so there is no driver clearing to wait out).  Labels are the context's own: 0 = the class the first
    // search's inputs landed in, 1 = "not 0" (not split further), 2 / 3 = the two other classes once a bulk search split them.
    std::vector<hipMemGenericAllocationHandle_t> io_spare[4];
    size_t io_spare_chunk = 0;                          // chunk size the spares were created with
    size_t io_spare_cap = 16;                           // chunks kept per class (IGDSP_IO_SPARE_CHUNKS; 0 = keep none)
    uint32_t io_epoch = 0;                              // moves with every full search: class labels of older buffer sets no longer apply
    std::mutex io_mu;
    bool is_spread(const void *p)
    {
        std::lock_guard<std::mutex> g(io_mu);
        for (const auto &r : spread_ranges) if ((const char *)p >= r.base && (const char *)p < r.base + r.bytes) return true;
        return false;
    }

    // a4 routing: direct table for 0 <= call_id < 65536 (lock-free reads), map beyond
    std::vector<std::atomic<uint32_t>> direct;
    std::unordered_map<int32_t, uint32_t> far;
    std::mutex far_mu;

    // staging (host pinned): a ring of kStageDepth frames per channel, ring[slot][c][256] + rlen / rpt / red per slot (slot-major: the
    // frames all calls staged at the same tick position sit next to each other, so the flush reads sequentially); head counts frames
    // written, tail frames taken by a flush (head - tail <= kStageDepth).  tp_adapter::payload_buff[256] semantics per slot
    // (TransportAdapter.h:66): the reference's hook runs on EVERY frame (TransportAdapter.cpp:303), so every frame is kept.
    // red = the call's ED-137 word when the frame was staged (igdsp_set_ed137 = setIncomingED137Value, roip_ed137.h:273).
    uint8_t *h_ring = nullptr;
    uint16_t *h_rlen = nullptr;
    uint8_t *h_rpt = nullptr;
    uint32_t *h_red = nullptr;
    std::vector<uint32_t> head, tail;
    std::vector<std::atomic<uint32_t>> cur_ed137;       // per channel: the word frames staged from now on carry
    std::vector<std::atomic_flag> slot_lock;
    std::atomic<uint32_t> hi_water{0};                 // 1 + highest channel ever staged
    std::atomic<uint32_t> gate_mode{IGDSP_GATE_ALWAYS}; // igdsp_set_gate_mode: how a flush gates each frame's fold into the window

    // flush: one pinned upload block + its device mirror (sections at 256-byte aligned offsets), per-channel device state and
    // its published host snapshots.  igdsp_flush_begin snapshots + enqueues, igdsp_flush_end waits and publishes; the device
    // writes each flush's {newest record, hold, probe} of every channel into the BACK set of pinned arrays, flush_end makes it
    // the front set (pub_seq moves): igdsp_poll / get_hold / get_probe read the front set and never take flush_mu.
    uint8_t *h_up = nullptr, *d_up = nullptr;
    size_t up_bytes = 0;
    igdsp_frame_stats *d_stats = nullptr;               // one record per staged frame of this flush (regions may have gaps, see flush_begin)
    igdsp_frame_stats *d_last = nullptr;                // per channel: record of its newest metered frame
    igdsp_chan_hold *d_hold = nullptr;
    igdsp_chan_probe *d_probe = nullptr;
    struct Published { igdsp_frame_stats *last; igdsp_chan_hold *hold; igdsp_chan_probe *probe; };
    Published pub[2] = {};                              // pinned; pub[front] is what the poll entries read
    std::atomic<uint32_t> front{0};
    std::atomic<uint64_t> pub_seq{0};                   // moves whenever `front` does (readers retry when it moved under them)
    std::vector<std::atomic<uint32_t>> frames_seen, frames_dropped;   // per channel: frames taken by a flush / overwritten before one took them
    hipEvent_t flush_done = nullptr;
    bool flush_open = false;                            // a flush_begin without its flush_end
    uint32_t flush_nch = 0;                             // channels the open flush downloads
    std::mutex flush_mu;                                // owner-thread entries (flush*, reset_hold)

    // snapshot helpers: at many channels the flush's host-side snapshot (one pass over every channel's ring) is shared out to a
    // few threads (igdsp_capi.hip: SnapshotPool)
    struct SnapshotPool;
    SnapshotPool *pool = nullptr;

    // device-wide work counters, ONE PAIR PER LAUNCH STREAM.  A persistent kernel draws its batches from the counter pair {next
    // batch, blocks done} it is handed and its last block re-arms the pair, so two kernels may share a pair only if they can never
    // be in flight together: launches on one stream serialise, launches on different streams do not.  (Round 2 handed launch k pair
    // k % 64 with no completion check: with >= 64 launches pending on one stream a launch on another stream received a pair that
    // was still in use — both kernels then drew batches from one counter: silently wrong records.)  A stream keeps its pair until
    // igdsp_sync(ctx, stream) has seen it idle; when all pairs are taken a launch on a new stream runs the static per-block
    // schedule (gqueue = nullptr: same results, ~2.5 % slower on the headline shape).
    uint32_t *d_queues = nullptr;                       // kQueueRing x 32 words (pair i at word 32 i: 128 B apart)
    // igdsp_encode's full 16-bit compressor table (2 laws x 65 536 codes) per encoder lineage, built once on the device by the
    // arithmetic every kernel shares (enc_uni) the first time a large batch asks for it; k_encode_lut16 then COPIES its 128 KiB
    // of LDS instead of evaluating 131 072 inputs per block and launch.  nullptr: not built (the kernel evaluates, as before).
    uint8_t *d_enc_tab[2] = {nullptr, nullptr};
    std::once_flag enc_once[2];
    hipStream_t queue_owner[64];                        // stream that owns pair i, or kFreeQueue (meaningful for i < queue_used)
    uint32_t queue_launches[64];                        // launches handed pair i since it was taken (igdsp_sync's idle check)
    uint32_t queue_used = 0;
    uint32_t queue_hint = 0;                            // pair of the previous launch: the common case is one stream
    std::atomic_flag queue_lk = ATOMIC_FLAG_INIT;
    bool global_queue = true;                           // device-wide batched work queue for k_meter_chunk64 (IGDSP_GLOBAL_QUEUE=0:
                                                        // static per-block batches).  Removes the inter-CU tail: -2.5 % on the
                                                        // headline launch once the outputs sit in another memory region than the
                                                        // payload (tools/kernel_ab.py); neutral when they share a region.
};
namespace {
constexpr uint32_t kQueueRing = 64;
const hipStream_t kFreeQueue = reinterpret_cast<hipStream_t>(~(uintptr_t)0);   // not a stream handle (NULL is one: the legacy default stream)
}

void igdsp_io_drop_spares(igdsp_ctx *ctx);             // igdsp_io.hip: give the spare chunks back (igdsp_destroy, and before a new search)

static inline int fail(igdsp_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        char buf[256];
        if (e != hipSuccess) std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else std::snprintf(buf, sizeof buf, "%s", what);
        ctx->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, call)                                                    \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) return fail((ctx), IGDSP_EDEVICE, #call, e_);   \
    } while (0)

// NULL means what it means everywhere in HIP: the legacy default (null) stream, so a caller that
// passes nothing stays ordered with its own default-stream work (e.g. torch tensors it just filled).
static inline hipStream_t pick(igdsp_ctx *, void *stream) { return (hipStream_t)stream; }
// The work-counter pair of `stream` (see igdsp_ctx::d_queues).  hipStreamPerThread names a different stream in every host thread,
// so it cannot key a pair: those launches take the static schedule.
static inline uint32_t *queue_of(igdsp_ctx *ctx, hipStream_t stream)
{
    if (!ctx->global_queue || !ctx->d_queues || stream == hipStreamPerThread) return nullptr;
    while (ctx->queue_lk.test_and_set(std::memory_order_acquire)) {}
    uint32_t i = ctx->queue_hint;
    if (i >= ctx->queue_used || ctx->queue_owner[i] != stream) {
        uint32_t free_i = kQueueRing;
        for (i = 0; i < ctx->queue_used && ctx->queue_owner[i] != stream; ++i)
            if (ctx->queue_owner[i] == kFreeQueue && free_i == kQueueRing) free_i = i;
        if (i == ctx->queue_used) {                      // a stream without a pair: a released position first, else one more
            if (free_i != kQueueRing) i = free_i;
            else if (ctx->queue_used < kQueueRing) ctx->queue_used += 1;
            else i = kQueueRing;                          // all taken: static schedule
            if (i < kQueueRing) { ctx->queue_owner[i] = stream; ctx->queue_launches[i] = 0; }
        }
    }
    uint32_t *q = nullptr;
    if (i < kQueueRing) { ctx->queue_hint = i; ctx->queue_launches[i] += 1; q = ctx->d_queues + 32u * i; }
    ctx->queue_lk.clear(std::memory_order_release);
    return q;
}
// igdsp_sync brackets hipStreamSynchronize(stream) with these two: if no launch took the stream's pair in between, nothing of the
// stream is in flight (its last kernel's last block has re-armed the pair) and the position is free for another stream.
static inline uint32_t queue_mark(igdsp_ctx *ctx, hipStream_t stream, uint32_t *pos)
{
    uint32_t n = 0;
    *pos = kQueueRing;
    while (ctx->queue_lk.test_and_set(std::memory_order_acquire)) {}
    for (uint32_t i = 0; i < ctx->queue_used; ++i)
        if (ctx->queue_owner[i] == stream) { *pos = i; n = ctx->queue_launches[i]; break; }
    ctx->queue_lk.clear(std::memory_order_release);
    return n;
}
static inline void queue_release_if_idle(igdsp_ctx *ctx, hipStream_t stream, uint32_t pos, uint32_t launches_before)
{
    if (pos >= kQueueRing) return;
    while (ctx->queue_lk.test_and_set(std::memory_order_acquire)) {}
    if (pos < ctx->queue_used && ctx->queue_owner[pos] == stream && ctx->queue_launches[pos] == launches_before) {
        ctx->queue_owner[pos] = kFreeQueue;
        while (ctx->queue_used && ctx->queue_owner[ctx->queue_used - 1] == kFreeQueue) ctx->queue_used -= 1;
    }
    ctx->queue_lk.clear(std::memory_order_release);
}
static inline igdsp::LaunchCfg cfg_of(igdsp_ctx *ctx, hipStream_t stream)
{
    return igdsp::LaunchCfg{ctx->cus, queue_of(ctx, stream)};
}

