// igdsp_capi.hip — the extern "C" boundary of include/igdsp.h over the gfx950
// kernels.  Host-side responsibilities only: context / stream / buffers, the
// single-frame staging slab behind setIncomingRTP/setOutgoingRTP
// (roip_ed137.cpp:6500-6587), call-id routing (roip_ed137.cpp:6519-6534) and
// argument validation.  There is NO CPU compute path here: when the HIP runtime
// or a gfx950 device is missing every entry fails with IGDSP_ENODEV.
#include "igdsp_ctx.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

using namespace igdsp;

namespace {
// Layout of the flush upload block (same offsets in the pinned host copy and in its device mirror; every section starts on
// a 256-byte boundary).  Group A = whole 160-byte frames, dense at stride 160 (the tuned chunk kernel's layout); group B =
// every other length, slots of 256 bytes with a length per frame (the general kernel).  seq = every staged frame in the
// order its channel received it: {record id (group B: | 0x80000000), ED-137 word}; runs[c] = {first seq entry, count} of
// channel c (count 0: nothing staged).
struct UploadLayout { size_t payA, payB, lenB, ptA, ptB, seq, runs, total; };
inline UploadLayout upload_layout(size_t max_frames, size_t max_channels)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    UploadLayout L;
    size_t o = 0;
    L.payA = o; o = up(o + max_frames * IGDSP_SAMPLES_PER_FRAME);
    L.payB = o; o = up(o + max_frames * kSlot);
    L.lenB = o; o = up(o + max_frames * sizeof(uint16_t));
    L.ptA = o;  o = up(o + max_frames);
    L.ptB = o;  o = up(o + max_frames);
    L.seq = o;  o = up(o + max_frames * 2 * sizeof(uint32_t));
    L.runs = o; o = up(o + max_channels * 2 * sizeof(uint32_t));
    L.total = o;
    return L;
}
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}
constexpr uint32_t kRecB = 0x80000000u;
constexpr uint32_t kPoolMinChannels = 16384;            // below this one thread snapshots faster than a pool wakes up
constexpr uint32_t kPoolMaxThreads = 16;               // (8 threads: 0.44-0.53 ms of owner time at 65 536 calls depending on the box; 16: below)

// What one snapshot worker found in its channel range [c0, c1): its frames sit in ITS region of every section (frame index
// c0 * kStageDepth onwards), so workers never touch each other's bytes.
struct SnapPart { uint32_t c0 = 0, c1 = 0, nA = 0, nB = 0, nSeq = 0; };
}  // namespace

// A few persistent helper threads for the flush's snapshot at many channels.  run() hands part i to thread i (the caller takes
// part 0) and returns when all are done.
struct igdsp_ctx::SnapshotPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t epoch = 0;
    uint32_t pending = 0;
    bool quit = false;
    std::function<void(uint32_t)> job;

    explicit SnapshotPool(uint32_t helpers)
    {
        for (uint32_t i = 0; i < helpers; ++i)
            threads.emplace_back([this, i] {
                uint64_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(m);
                    cv_go.wait(lk, [&] { return quit || epoch != seen; });
                    if (quit) return;
                    seen = epoch;
                    lk.unlock();
                    job(i + 1);
                    lk.lock();
                    if (--pending == 0) cv_done.notify_one();
                }
            });
    }
    ~SnapshotPool()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_go.notify_all();
        for (auto &t : threads) t.join();
    }
    void run(const std::function<void(uint32_t)> &fn)
    {
        { std::lock_guard<std::mutex> lk(m); job = fn; pending = (uint32_t)threads.size(); ++epoch; }
        cv_go.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

// Read one channel's published state: copy out of the front set, retry if a flush_end moved it meanwhile (two flushes would
// have to complete within the copy of ~60 bytes for a second retry).
template <typename Fn>
static inline void read_published(igdsp_ctx *ctx, Fn &&fn)
{
    for (;;) {
        const uint64_t s1 = ctx->pub_seq.load(std::memory_order_acquire);
        if (s1 & 1u) { cpu_relax(); continue; }                       // igdsp_reset_hold is rewriting the front set
        fn(ctx->pub[ctx->front.load(std::memory_order_acquire)]);
        std::atomic_thread_fence(std::memory_order_acquire);
        if (ctx->pub_seq.load(std::memory_order_relaxed) == s1) return;
    }
}

extern "C" {

int igdsp_abi_version(void) { return IGDSP_ABI_VERSION; }

int igdsp_create(igdsp_ctx **out, int device, uint32_t max_channels)
{
    if (!out || max_channels == 0 || max_channels > (1u << 24)) return IGDSP_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return IGDSP_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return IGDSP_ENODEV;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return IGDSP_ENODEV;   // code objects are gfx950-only
    if (hipSetDevice(device) != hipSuccess) return IGDSP_ENODEV;

    igdsp_ctx *ctx = new (std::nothrow) igdsp_ctx();
    if (!ctx) return IGDSP_ENOMEM;
    ctx->device = device;
    ctx->cus = prop.multiProcessorCount;
    ctx->name = prop.name;
    ctx->max_channels = max_channels;
    ctx->direct = std::vector<std::atomic<uint32_t>>(kDirectCalls);
    for (auto &d : ctx->direct) d.store(kNoChan, std::memory_order_relaxed);
    ctx->slot_lock = std::vector<std::atomic_flag>(max_channels);
    for (auto &f : ctx->slot_lock) f.clear();
    ctx->cur_ed137 = std::vector<std::atomic<uint32_t>>(max_channels);
    ctx->frames_seen = std::vector<std::atomic<uint32_t>>(max_channels);
    ctx->frames_dropped = std::vector<std::atomic<uint32_t>>(max_channels);
    for (uint32_t c = 0; c < max_channels; ++c) { ctx->cur_ed137[c].store(0); ctx->frames_seen[c].store(0); ctx->frames_dropped[c].store(0); }
    ctx->head.assign(max_channels, 0);
    ctx->tail.assign(max_channels, 0);
    const size_t max_frames = (size_t)max_channels * kStageDepth;       // most frames one flush can take
    const size_t ring = max_frames * kSlot;
    ctx->up_bytes = upload_layout(max_frames, max_channels).total;
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->flush_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->h_ring, ring, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->h_rlen, max_frames * sizeof(uint16_t), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->h_rpt, max_frames, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->h_red, max_frames * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->h_up, ctx->up_bytes, hipHostMallocDefault) == hipSuccess;
    for (auto &pb : ctx->pub) {
        ok = ok && hipHostMalloc((void **)&pb.last, max_channels * sizeof(igdsp_frame_stats), hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&pb.hold, max_channels * sizeof(igdsp_chan_hold), hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&pb.probe, max_channels * sizeof(igdsp_chan_probe), hipHostMallocDefault) == hipSuccess;
    }
    ok = ok && hipMalloc((void **)&ctx->d_up, ctx->up_bytes) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_stats, 2 * max_frames * sizeof(igdsp_frame_stats)) == hipSuccess;   // group A | group B
    ok = ok && hipMalloc((void **)&ctx->d_last, max_channels * sizeof(igdsp_frame_stats)) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_hold, max_channels * sizeof(igdsp_chan_hold)) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_probe, max_channels * sizeof(igdsp_chan_probe)) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_queues, kQueueRing * 32u * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(ctx->d_queues, 0, kQueueRing * 32u * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(ctx->d_last, 0, max_channels * sizeof(igdsp_frame_stats)) == hipSuccess;
    ok = ok && hipMemset(ctx->d_probe, 0, max_channels * sizeof(igdsp_chan_probe)) == hipSuccess;
    if (const char *e = std::getenv("IGDSP_GLOBAL_QUEUE")) ctx->global_queue = std::atoi(e) != 0;
    if (!ok) {
        igdsp_destroy(ctx);
        return IGDSP_ENOMEM;
    }
    if (init_device_attributes() != hipSuccess) {      // this device's kernel attributes (hipSetDevice above)
        igdsp_destroy(ctx);
        return IGDSP_EDEVICE;
    }
    bool up = launch_hold_reset(ctx->d_hold, max_channels, nullptr, ctx->stream) == hipSuccess;
    for (auto &pb : ctx->pub) {
        std::memset(pb.last, 0, max_channels * sizeof(igdsp_frame_stats));
        std::memset(pb.probe, 0, max_channels * sizeof(igdsp_chan_probe));
        up = up && hipMemcpyAsync(pb.hold, ctx->d_hold, max_channels * sizeof(igdsp_chan_hold), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    }
    if (!up || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        igdsp_destroy(ctx);
        return IGDSP_EDEVICE;
    }
    if (max_channels >= kPoolMinChannels) {
        const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
        uint32_t threads = std::min(std::min(kPoolMaxThreads, std::max(1u, hw / 2u)), max_channels / (kPoolMinChannels / 4u));   // >= 4 096 channels per thread
        if (const char *e = std::getenv("IGDSP_FLUSH_THREADS")) threads = (uint32_t)std::max(1, std::min(64, std::atoi(e)));
        if (threads > 1u) ctx->pool = new (std::nothrow) igdsp_ctx::SnapshotPool(threads - 1u);
    }
    *out = ctx;
    return IGDSP_OK;
}

int igdsp_destroy(igdsp_ctx *ctx)
{
    if (!ctx) return IGDSP_OK;                       // tolerate NULL like the reference's setters (TransportAdapter.cpp:135-223)
    delete ctx->pool;
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    igdsp_io_drop_spares(ctx);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->flush_done) (void)hipEventDestroy(ctx->flush_done);
    void *hosts[] = {ctx->h_ring, ctx->h_rlen, ctx->h_rpt, ctx->h_red, ctx->h_up, ctx->pub[0].last, ctx->pub[0].hold, ctx->pub[0].probe,
                     ctx->pub[1].last, ctx->pub[1].hold, ctx->pub[1].probe};
    for (void *p : hosts) if (p) (void)hipHostFree(p);
    void *devs[] = {ctx->d_up, ctx->d_stats, ctx->d_last, ctx->d_hold, ctx->d_probe, ctx->d_queues, ctx->d_enc_tab[0], ctx->d_enc_tab[1]};
    for (void *p : devs) if (p) (void)hipFree(p);
    delete ctx;
    return IGDSP_OK;
}

const char *igdsp_last_error(const igdsp_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int igdsp_device_info(const igdsp_ctx *ctx, int *device, int *compute_units, char *name, size_t name_len)
{
    if (!ctx) return IGDSP_EINVAL;
    if (device) *device = ctx->device;
    if (compute_units) *compute_units = ctx->cus;
    if (name && name_len) { std::strncpy(name, ctx->name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    return IGDSP_OK;
}

int igdsp_set_variant(igdsp_ctx *ctx, int variant)
{
    if (!ctx || variant < 0 || variant > 4) return IGDSP_EINVAL;
    ctx->variant = variant;
    return IGDSP_OK;
}

// ---------------------------------------------------------------- routing (a4)
static uint32_t lookup(igdsp_ctx *ctx, int32_t call_id)
{
    if (call_id >= 0 && call_id < kDirectCalls) return ctx->direct[(size_t)call_id].load(std::memory_order_acquire);
    std::lock_guard<std::mutex> g(ctx->far_mu);
    auto it = ctx->far.find(call_id);
    return it == ctx->far.end() ? kNoChan : it->second;
}

int igdsp_map_call(igdsp_ctx *ctx, int32_t call_id, uint32_t channel)
{
    if (!ctx) return IGDSP_EINVAL;
    if (channel >= ctx->max_channels) return IGDSP_ERANGE;
    if (call_id >= 0 && call_id < kDirectCalls) ctx->direct[(size_t)call_id].store(channel, std::memory_order_release);
    else { std::lock_guard<std::mutex> g(ctx->far_mu); ctx->far[call_id] = channel; }
    return IGDSP_OK;
}

int igdsp_unmap_call(igdsp_ctx *ctx, int32_t call_id)
{
    if (!ctx) return IGDSP_EINVAL;
    if (call_id >= 0 && call_id < kDirectCalls) ctx->direct[(size_t)call_id].store(kNoChan, std::memory_order_release);
    else { std::lock_guard<std::mutex> g(ctx->far_mu); ctx->far.erase(call_id); }
    return IGDSP_OK;
}

// ---------------------------------------------------------------- single-frame entry
int igdsp_on_rtp_frame(igdsp_ctx *ctx, int32_t call_id, uint8_t pt, const uint8_t *payload, uint32_t payloadlen)
{
    if (!ctx) return IGDSP_EINVAL;
    if (pt != IGDSP_PT_PCMU && pt != IGDSP_PT_PCMA) return IGDSP_OK;   // keep-alive (123) / other codecs: not metered
    if (payloadlen > kSlot || (payloadlen && !payload)) return IGDSP_EINVAL;
    const uint32_t ch = lookup(ctx, call_id);
    if (ch == kNoChan) return IGDSP_ENOENT;          // the reference's if-chain falls through silently; we report it
    if (payloadlen == 0) return IGDSP_OK;
    std::atomic_flag &lk = ctx->slot_lock[ch];
    while (lk.test_and_set(std::memory_order_acquire)) cpu_relax();     // held by another producer / the flush for one <= 256-byte copy
    int rc = IGDSP_OK;
    if (ctx->head[ch] - ctx->tail[ch] == kStageDepth) {                 // the owner thread is > 160 ms late: the oldest frame goes
        ctx->tail[ch] += 1;
        ctx->frames_dropped[ch].fetch_add(1, std::memory_order_relaxed);
        rc = IGDSP_EBUSY;
    }
    const size_t slot = (size_t)(ctx->head[ch] % kStageDepth) * ctx->max_channels + ch;   // slot-major: the flush walks each slot plane sequentially
    std::memcpy(ctx->h_ring + slot * kSlot, payload, payloadlen);
    ctx->h_rlen[slot] = (uint16_t)payloadlen;
    ctx->h_rpt[slot] = pt;
    ctx->h_red[slot] = ctx->cur_ed137[ch].load(std::memory_order_relaxed);
    ctx->head[ch] += 1;
    lk.clear(std::memory_order_release);
    uint32_t hw = ctx->hi_water.load(std::memory_order_relaxed);
    while (hw < ch + 1 && !ctx->hi_water.compare_exchange_weak(hw, ch + 1, std::memory_order_relaxed)) {}
    return rc;
}

// setIncomingED137Value (roip_ed137.h:273): the word the call's frames carry from now on
int igdsp_set_ed137(igdsp_ctx *ctx, int32_t call_id, uint32_t ed137_value)
{
    if (!ctx) return IGDSP_EINVAL;
    const uint32_t ch = lookup(ctx, call_id);
    if (ch == kNoChan) return IGDSP_ENOENT;
    ctx->cur_ed137[ch].store(ed137_value, std::memory_order_relaxed);
    return IGDSP_OK;
}

int igdsp_set_gate_mode(igdsp_ctx *ctx, uint32_t gate_mode)
{
    if (!ctx || gate_mode > IGDSP_GATE_SQU_OR_PTT) return IGDSP_EINVAL;
    ctx->gate_mode.store(gate_mode, std::memory_order_relaxed);
    return IGDSP_OK;
}

// One worker's share of the snapshot: every channel of [part.c0, part.c1), staged frames oldest first, compacted into the
// worker's own region of the upload block, under the channel's flag.
static void snapshot_part(igdsp_ctx *ctx, const UploadLayout &L, SnapPart &part)
{
    uint8_t *up = ctx->h_up;
    uint16_t *lenB = reinterpret_cast<uint16_t *>(up + L.lenB);
    uint32_t *seq = reinterpret_cast<uint32_t *>(up + L.seq), *runs = reinterpret_cast<uint32_t *>(up + L.runs);
    const uint32_t base = part.c0 * kStageDepth;                       // first frame index of this worker's regions
    uint32_t nA = 0, nB = 0, nS = 0;
    for (uint32_t c = part.c0; c < part.c1; ++c) {
        std::atomic_flag &lk = ctx->slot_lock[c];
        while (lk.test_and_set(std::memory_order_acquire)) cpu_relax();
        const uint32_t t0 = ctx->tail[c], h0 = ctx->head[c];
        const uint32_t s0 = nS;
        for (uint32_t k = t0; k != h0; ++k) {
            const size_t slot = (size_t)(k % kStageDepth) * ctx->max_channels + c;
            const uint16_t l = ctx->h_rlen[slot];
            uint32_t id;
            if (l == IGDSP_SAMPLES_PER_FRAME) {
                id = base + nA++;
                std::memcpy(up + L.payA + (size_t)id * IGDSP_SAMPLES_PER_FRAME, ctx->h_ring + slot * kSlot, l);
                up[L.ptA + id] = ctx->h_rpt[slot];
            } else {
                const uint32_t ib = base + nB++;
                std::memcpy(up + L.payB + (size_t)ib * kSlot, ctx->h_ring + slot * kSlot, l);
                lenB[ib] = l;
                up[L.ptB + ib] = ctx->h_rpt[slot];
                id = ib | kRecB;
            }
            seq[2 * (size_t)(base + nS)] = id;
            seq[2 * (size_t)(base + nS) + 1] = ctx->h_red[slot];
            ++nS;
        }
        ctx->tail[c] = h0;
        lk.clear(std::memory_order_release);
        runs[2 * (size_t)c] = base + s0;
        runs[2 * (size_t)c + 1] = nS - s0;
        if (h0 != t0) ctx->frames_seen[c].fetch_add(h0 - t0, std::memory_order_relaxed);
    }
    part.nA = nA; part.nB = nB; part.nSeq = nS;
}

static int flush_end_locked(igdsp_ctx *ctx, int wait)
{
    if (!ctx->flush_open) return IGDSP_OK;
    if (!wait) {
        const hipError_t q = hipEventQuery(ctx->flush_done);
        if (q == hipErrorNotReady) return IGDSP_EBUSY;
        if (q != hipSuccess) return fail(ctx, IGDSP_EDEVICE, "hipEventQuery(flush_done)", q);
    } else {
        HIP_TRY(ctx, hipEventSynchronize(ctx->flush_done));
    }
    // the back set is complete: make it the front set.  Readers that were half way through the old one notice pub_seq moving.
    ctx->front.store(ctx->front.load(std::memory_order_relaxed) ^ 1u, std::memory_order_release);
    ctx->pub_seq.fetch_add(2, std::memory_order_release);
    ctx->flush_open = false;
    return IGDSP_OK;
}

static int flush_begin_locked(igdsp_ctx *ctx, uint32_t *n_frames_out)
{
    if (int rc = flush_end_locked(ctx, 1)) return rc;               // one flush at a time: the upload block is single
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t nch = ctx->hi_water.load(std::memory_order_relaxed);
    if (n_frames_out) *n_frames_out = 0;
    if (nch == 0) return IGDSP_OK;
    const size_t max_frames = (size_t)ctx->max_channels * kStageDepth;
    const UploadLayout L = upload_layout(max_frames, ctx->max_channels);
    // 1. snapshot every channel's staged frames (oldest first) into the upload block, compacted per worker region
    SnapPart parts[64];
    uint32_t n_parts = 1;
    if (ctx->pool && nch >= kPoolMinChannels) n_parts = std::min<uint32_t>((uint32_t)ctx->pool->threads.size() + 1u, 64u);
    for (uint32_t i = 0; i < n_parts; ++i) { parts[i].c0 = (uint32_t)((uint64_t)nch * i / n_parts); parts[i].c1 = (uint32_t)((uint64_t)nch * (i + 1) / n_parts); }
    if (n_parts == 1) snapshot_part(ctx, L, parts[0]);
    else ctx->pool->run([&](uint32_t i) { if (i < n_parts) snapshot_part(ctx, L, parts[i]); });
    uint32_t staged = 0, endA = 0, endB = 0, endS = 0;
    for (uint32_t i = 0; i < n_parts; ++i) {
        staged += parts[i].nSeq;
        if (parts[i].nA) endA = parts[i].c0 * kStageDepth + parts[i].nA;
        if (parts[i].nB) endB = parts[i].c0 * kStageDepth + parts[i].nB;
        if (parts[i].nSeq) endS = parts[i].c0 * kStageDepth + parts[i].nSeq;
    }
    if (n_frames_out) *n_frames_out = staged;
    if (staged == 0) return IGDSP_OK;
    // 2. upload what is used: the payload regions per worker (the big ones), the small sections as one span each; meter group A
    //    and group B over their spans (frames between two workers' regions are stale bytes: their records are never looked at),
    //    fold every channel's frames in arrival order, download the per-channel state into the back set
    hipStream_t s = ctx->stream;
    uint8_t *up = ctx->h_up, *d = ctx->d_up;
    auto copy = [&](size_t off, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(d + off, up + off, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
    };
    for (uint32_t i = 0; i < n_parts; ++i) {
        const size_t base = (size_t)parts[i].c0 * kStageDepth;
        HIP_TRY(ctx, copy(L.payA + base * IGDSP_SAMPLES_PER_FRAME, (size_t)parts[i].nA * IGDSP_SAMPLES_PER_FRAME));
        HIP_TRY(ctx, copy(L.payB + base * kSlot, (size_t)parts[i].nB * kSlot));
    }
    HIP_TRY(ctx, copy(L.ptA, endA));
    HIP_TRY(ctx, copy(L.ptB, endB));
    HIP_TRY(ctx, copy(L.lenB, (size_t)endB * sizeof(uint16_t)));
    HIP_TRY(ctx, copy(L.seq, (size_t)endS * 2 * sizeof(uint32_t)));
    HIP_TRY(ctx, copy(L.runs, (size_t)nch * 2 * sizeof(uint32_t)));
    // records: group A's at d_stats[id], group B's at d_stats[max_frames + id] (ids are region-based, so each group may reach max_frames)
    igdsp_frame_stats *stA = ctx->d_stats, *stB = ctx->d_stats + max_frames;
    if (endA)   // whole 160-byte frames, dense: the chunk kernel takes every 64, the general kernel the < 64 left over
        HIP_TRY(ctx, launch_decode_meter(cfg_of(ctx, s), 0, d + L.payA, d + L.ptA, nullptr, endA, 1, IGDSP_SAMPLES_PER_FRAME, stA, nullptr, nullptr, 0, s));
    for (uint32_t i = 0; i < n_parts; ++i)   // every other length (rare): 256-byte slots with a length per frame, one launch per region that has any
        if (parts[i].nB) {
            const size_t base = (size_t)parts[i].c0 * kStageDepth;
            HIP_TRY(ctx, launch_decode_meter(cfg_of(ctx, s), 1, d + L.payB + base * kSlot, d + L.ptB + base, reinterpret_cast<const uint16_t *>(d + L.lenB) + base,
                                             parts[i].nB, 1, kSlot, stB + base, nullptr, nullptr, 0, s));
        }
    HIP_TRY(ctx, launch_flush_fold(stA, stB, reinterpret_cast<const uint16_t *>(d + L.lenB), reinterpret_cast<const uint2 *>(d + L.seq),
                                   reinterpret_cast<const uint2 *>(d + L.runs), nch, ctx->gate_mode.load(std::memory_order_relaxed), IGDSP_PROBE_ALARM,
                                   ctx->d_hold, ctx->d_probe, ctx->d_last, s));
    const igdsp_ctx::Published &back = ctx->pub[ctx->front.load(std::memory_order_relaxed) ^ 1u];
    HIP_TRY(ctx, hipMemcpyAsync(back.last, ctx->d_last, (size_t)nch * sizeof(igdsp_frame_stats), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(back.hold, ctx->d_hold, (size_t)nch * sizeof(igdsp_chan_hold), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(back.probe, ctx->d_probe, (size_t)nch * sizeof(igdsp_chan_probe), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(ctx->flush_done, s));
    ctx->flush_open = true;
    ctx->flush_nch = nch;
    return IGDSP_OK;
}

int igdsp_flush_begin(igdsp_ctx *ctx, uint32_t *n_frames_out)
{
    if (!ctx) return IGDSP_EINVAL;
    std::lock_guard<std::mutex> g(ctx->flush_mu);
    return flush_begin_locked(ctx, n_frames_out);
}

int igdsp_flush_end(igdsp_ctx *ctx, int wait)
{
    if (!ctx) return IGDSP_EINVAL;
    std::lock_guard<std::mutex> g(ctx->flush_mu);
    return flush_end_locked(ctx, wait);
}

int igdsp_flush(igdsp_ctx *ctx, uint32_t *n_frames_out)
{
    if (!ctx) return IGDSP_EINVAL;
    std::lock_guard<std::mutex> g(ctx->flush_mu);
    if (int rc = flush_begin_locked(ctx, n_frames_out)) return rc;
    return flush_end_locked(ctx, 1);
}

int igdsp_poll(igdsp_ctx *ctx, uint32_t channel, igdsp_level *out)
{
    if (!ctx || !out) return IGDSP_EINVAL;
    if (channel >= ctx->max_channels) return IGDSP_ERANGE;
    igdsp_frame_stats s;
    uint16_t peak_hold = 0;
    read_published(ctx, [&](const igdsp_ctx::Published &p) { s = p.last[channel]; peak_hold = p.hold[channel].peak_hold; });
    out->byte_mean = s.byte_mean;
    out->flags = s.flags;
    out->peak = s.peak;
    out->rms = s.rms;
    out->percent = (int32_t)(float)(((double)s.rms * 100.0) / (double)IGDSP_METER_FULL_SCALE);   // audiometer.cpp:30-31
    out->peak_hold = peak_hold;
    out->dropped = (uint16_t)std::min<uint32_t>(ctx->frames_dropped[channel].load(std::memory_order_relaxed), 65535u);
    out->frames = ctx->frames_seen[channel].load(std::memory_order_relaxed);
    return IGDSP_OK;
}

int igdsp_poll_call(igdsp_ctx *ctx, int32_t call_id, igdsp_level *out)
{
    if (!ctx || !out) return IGDSP_EINVAL;
    const uint32_t ch = lookup(ctx, call_id);
    if (ch == kNoChan) return IGDSP_ENOENT;
    return igdsp_poll(ctx, ch, out);
}

int igdsp_reset_hold(igdsp_ctx *ctx, uint32_t channel)
{
    if (!ctx) return IGDSP_EINVAL;
    if (channel != 0xFFFFFFFFu && channel >= ctx->max_channels) return IGDSP_ERANGE;
    std::lock_guard<std::mutex> g(ctx->flush_mu);
    if (int rc = flush_end_locked(ctx, 1)) return rc;                  // a flush under way folds into the window being reset: finish it first
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t c0 = (channel == 0xFFFFFFFFu) ? 0 : channel;
    const uint32_t cn = (channel == 0xFFFFFFFFu) ? ctx->max_channels : 1;
    HIP_TRY(ctx, launch_hold_reset(ctx->d_hold + c0, cn, nullptr, ctx->stream));
    // both published sets show the reset window at once (no flush is open, so nothing else writes them)
    const uint32_t f = ctx->front.load(std::memory_order_relaxed);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pub[f ^ 1u].hold + c0, ctx->d_hold + c0, cn * sizeof(igdsp_chan_hold), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pub_seq.fetch_add(1, std::memory_order_acq_rel);             // odd: readers wait
    std::memcpy(ctx->pub[f].hold + c0, ctx->pub[f ^ 1u].hold + c0, cn * sizeof(igdsp_chan_hold));
    ctx->pub_seq.fetch_add(1, std::memory_order_release);
    return IGDSP_OK;
}

int igdsp_get_hold(igdsp_ctx *ctx, uint32_t channel, igdsp_chan_hold *out)
{
    if (!ctx || !out) return IGDSP_EINVAL;
    if (channel >= ctx->max_channels) return IGDSP_ERANGE;
    read_published(ctx, [&](const igdsp_ctx::Published &p) { *out = p.hold[channel]; });
    return IGDSP_OK;
}

int igdsp_get_probe(igdsp_ctx *ctx, uint32_t channel, igdsp_chan_probe *out)
{
    if (!ctx || !out) return IGDSP_EINVAL;
    if (channel >= ctx->max_channels) return IGDSP_ERANGE;
    read_published(ctx, [&](const igdsp_ctx::Published &p) { *out = p.probe[channel]; });
    return IGDSP_OK;
}

// ---------------------------------------------------------------- batched device entries
static int check_shape(uint32_t C, uint32_t F, uint32_t n)
{
    if (n == 0 || n > IGDSP_MAX_PAYLOAD) return IGDSP_EINVAL;
    if ((uint64_t)C * F >= 0xFFFFFFE0ull) return IGDSP_ERANGE;     // frame indices are 32-bit on the device
    return IGDSP_OK;
}

int igdsp_decode_meter(igdsp_ctx *ctx, const uint8_t *d_payload, const uint8_t *d_codec, const uint16_t *d_len,
                       uint32_t C, uint32_t F, uint32_t n, igdsp_frame_stats *d_stats, int16_t *d_pcm,
                       igdsp_aggregate *d_agg, uint32_t rank, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;                      // empty batch: nothing to do
    if (!d_payload || !d_codec || !d_stats) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, n)) return rc;
    if (rank >= IGDSP_AGG_MAX_RANKS) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_decode_meter(cfg_of(ctx, pick(ctx, stream)), ctx->variant, d_payload, d_codec, d_len, C, F, n, d_stats, d_pcm, d_agg, rank, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_encode(igdsp_ctx *ctx, const int16_t *d_pcm, const uint8_t *d_codec, uint32_t C, uint32_t F, uint32_t n,
                 uint8_t *d_out, int variant, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_pcm || !d_codec || !d_out) return IGDSP_EINVAL;
    if (variant != IGDSP_ENC_SUN16 && variant != IGDSP_ENC_G191) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, n)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    igdsp::LaunchCfg cfg = cfg_of(ctx, pick(ctx, stream));
    if ((uint64_t)C * F * n >= (1ull << 25)) {                     // the batches k_encode_lut16 serves: its table, built once per context and lineage
        const int v = variant == IGDSP_ENC_G191 ? 1 : 0;
        std::call_once(ctx->enc_once[v], [&]() {
            uint8_t *t = nullptr;
            hipStream_t bs = nullptr;
            bool ok = hipMalloc((void **)&t, 2u * 65536u) == hipSuccess && hipStreamCreateWithFlags(&bs, hipStreamNonBlocking) == hipSuccess;
            ok = ok && igdsp::launch_build_enc_table(variant, t, bs) == hipSuccess && hipStreamSynchronize(bs) == hipSuccess;
            if (bs) (void)hipStreamDestroy(bs);
            if (ok) ctx->d_enc_tab[v] = t;                          // complete before any launch that reads it is enqueued
            else { if (t) (void)hipFree(t); (void)hipGetLastError(); }   // the kernel evaluates the table itself, as before
        });
        cfg.enc_tab = ctx->d_enc_tab[v];
    }
    HIP_TRY(ctx, launch_encode(cfg, d_pcm, d_codec, C, F, n, d_out, variant, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_roundtrip_peakhold(igdsp_ctx *ctx, const uint8_t *d_payload, const uint8_t *d_codec, uint32_t C, uint32_t F,
                             uint32_t n, uint8_t *d_out, igdsp_frame_stats *d_stats, igdsp_chan_hold *d_hold,
                             const uint8_t *d_gate, int variant, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_payload || !d_codec || !d_out || !d_stats || !d_hold) return IGDSP_EINVAL;
    if (variant != IGDSP_ENC_SUN16 && variant != IGDSP_ENC_G191) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, n)) return rc;
    if ((reinterpret_cast<uintptr_t>(d_stats) & 7u) || (reinterpret_cast<uintptr_t>(d_hold) & 7u)) return IGDSP_EINVAL;   // natural struct alignment
    // every shape is served: whole groups of 64 channels of 160-byte frames by the fused channel-group-major kernel,
    // the remaining channels and every other geometry by the general wave-per-channel kernel (launch_roundtrip)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    igdsp::LaunchCfg cfg = cfg_of(ctx, pick(ctx, stream));
    cfg.out_spread = ctx->is_spread(d_out);
    HIP_TRY(ctx, launch_roundtrip(cfg, ctx->variant, d_payload, d_codec, C, F, n, d_out, d_stats, d_hold, d_gate, variant, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_hold_update(igdsp_ctx *ctx, const igdsp_frame_stats *d_stats, uint32_t C, uint32_t F, uint32_t n,
                      igdsp_chan_hold *d_hold, const uint8_t *d_gate, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_stats || !d_hold) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, n)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_hold_update(d_stats, nullptr, C, F, n, d_hold, d_gate, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_hold_reset(igdsp_ctx *ctx, igdsp_chan_hold *d_hold, uint32_t C, const uint8_t *d_mask, void *stream)
{
    if (!ctx || (!d_hold && C)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_hold_reset(d_hold, C, d_mask, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_agg_reset(igdsp_ctx *ctx, igdsp_aggregate *d_agg, void *stream)
{
    if (!ctx || !d_agg) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(d_agg, 0, sizeof(igdsp_aggregate), pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_depayload(igdsp_ctx *ctx, const uint8_t *d_packets, const uint16_t *d_sizes, const uint8_t *d_radio, uint32_t C,
                    uint32_t F, uint32_t pkt_stride, uint32_t n, uint8_t *d_payload_out, uint16_t *d_len_out,
                    igdsp_rtp_info *d_info_out, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_packets || !d_radio || !d_payload_out || !d_len_out || !d_info_out) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, n)) return rc;
    // slots hold at least a 20-byte header, are dword-granular (so header words and payload dwords are aligned)
    if (pkt_stride < 20u || (pkt_stride & 3u) || pkt_stride > 2048u || (reinterpret_cast<uintptr_t>(d_packets) & 3u)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_depayload(cfg_of(ctx, pick(ctx, stream)), d_packets, d_sizes, d_radio, C, F, pkt_stride, n, d_payload_out, d_len_out, d_info_out, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_decode_meter_rtp(igdsp_ctx *ctx, const uint8_t *d_slots, const uint8_t *d_codec, uint32_t C, uint32_t F,
                           igdsp_frame_stats *d_stats, igdsp_rtp_info *d_info, igdsp_aggregate *d_agg, uint32_t rank, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_slots || !d_codec || !d_stats || rank >= IGDSP_AGG_MAX_RANKS) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, IGDSP_SAMPLES_PER_FRAME)) return rc;
    // the fused kernel consumes whole 64-slot super-chunks of 16-byte aligned slots; other shapes take the
    // two-step route (igdsp_depayload + igdsp_decode_meter) — rejected here rather than silently re-routed
    if (((uint64_t)C * F) % 64u || (reinterpret_cast<uintptr_t>(d_slots) & 15u) || (reinterpret_cast<uintptr_t>(d_stats) & 15u) ||
        (reinterpret_cast<uintptr_t>(d_info) & 7u))
        return fail(ctx, IGDSP_EINVAL, "decode_meter_rtp needs C*F % 64 == 0 and 16-byte aligned slots / stats");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_decode_meter_rtp(cfg_of(ctx, pick(ctx, stream)), d_slots, nullptr, d_codec, C, F, 0, 20, d_stats, d_info, d_agg, rank, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_decode_meter_packets(igdsp_ctx *ctx, const uint8_t *d_packets, const uint16_t *d_sizes, const uint8_t *d_codec, uint32_t C,
                               uint32_t F, uint32_t pkt_stride, uint32_t hdr_bytes, igdsp_frame_stats *d_stats,
                               igdsp_rtp_info *d_info, igdsp_aggregate *d_agg, uint32_t rank, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_packets || !d_codec || !d_stats || rank >= IGDSP_AGG_MAX_RANKS) return IGDSP_EINVAL;
    if (hdr_bytes != 12u && hdr_bytes != 20u) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, IGDSP_SAMPLES_PER_FRAME)) return rc;
    if (pkt_stride < hdr_bytes + IGDSP_SAMPLES_PER_FRAME || pkt_stride < 20u || (pkt_stride & 3u) || pkt_stride > 2048u ||
        (uint64_t)C * F * pkt_stride > 0xFFFFFFFFull * 4ull)
        return IGDSP_EINVAL;
    if (((uint64_t)C * F) % 64u || (reinterpret_cast<uintptr_t>(d_packets) & 3u) || (reinterpret_cast<uintptr_t>(d_stats) & 15u) ||
        (reinterpret_cast<uintptr_t>(d_info) & 7u) || (reinterpret_cast<uintptr_t>(d_sizes) & 1u))
        return fail(ctx, IGDSP_EINVAL, "decode_meter_packets needs C*F % 64 == 0, dword-aligned packets, 16-byte aligned stats");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_decode_meter_rtp(cfg_of(ctx, pick(ctx, stream)), d_packets, d_sizes, d_codec, C, F, pkt_stride, hdr_bytes, d_stats, d_info, d_agg, rank, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_decode_meter_packets_mixed(igdsp_ctx *ctx, const uint8_t *d_packets, const uint16_t *d_sizes, const uint8_t *d_codec,
                                     const uint8_t *d_radio, uint32_t C, uint32_t F, uint32_t pkt_stride, igdsp_frame_stats *d_stats,
                                     igdsp_rtp_info *d_info, igdsp_aggregate *d_agg, uint32_t rank, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_packets || !d_codec || !d_radio || !d_stats || rank >= IGDSP_AGG_MAX_RANKS) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, IGDSP_SAMPLES_PER_FRAME)) return rc;
    if (pkt_stride < 180u || (pkt_stride & 3u) || pkt_stride > 2048u || (uint64_t)C * F * pkt_stride > 0xFFFFFFFFull * 4ull)
        return IGDSP_EINVAL;
    if (((uint64_t)C * F) % 64u || (reinterpret_cast<uintptr_t>(d_packets) & 3u) || (reinterpret_cast<uintptr_t>(d_stats) & 15u) ||
        (reinterpret_cast<uintptr_t>(d_info) & 7u) || (reinterpret_cast<uintptr_t>(d_sizes) & 1u))
        return fail(ctx, IGDSP_EINVAL, "decode_meter_packets_mixed needs C*F % 64 == 0, dword-aligned packets, 16-byte aligned stats");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_decode_meter_rtp(cfg_of(ctx, pick(ctx, stream)), d_packets, d_sizes, d_codec, C, F, pkt_stride, 12, d_stats, d_info, d_agg, rank, pick(ctx, stream), d_radio));
    return IGDSP_OK;
}

// ---------------------------------------------------------------- ED-137 gated window (last clause)
size_t igdsp_window_work_bytes(uint32_t n_channels) { return (size_t)kWinMaxSeg * 3u * n_channels * sizeof(uint4); }

static int check_window(igdsp_ctx *ctx, const igdsp_window *win)
{
    if (!win || !win->d_hold || win->gate_mode > IGDSP_GATE_SQU_OR_PTT) return IGDSP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(win->d_hold) & 7u) || (reinterpret_cast<uintptr_t>(win->d_probe) & 3u) || (reinterpret_cast<uintptr_t>(win->d_work) & 15u))
        return fail(ctx, IGDSP_EINVAL, "igdsp_window: d_hold 8-byte, d_probe 4-byte, d_work 16-byte aligned");
    return IGDSP_OK;
}

int igdsp_window_update(igdsp_ctx *ctx, const igdsp_frame_stats *d_stats, const igdsp_rtp_info *d_info, const uint16_t *d_len,
                        uint32_t C, uint32_t F, uint32_t n, const igdsp_window *win, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if (int rc = check_window(ctx, win)) return rc;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_stats) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, n)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_window_update(d_stats, d_info, d_len, C, F, n, win->gate_mode, win->probe_alarm ? win->probe_alarm : IGDSP_PROBE_ALARM,
                                      win->d_hold, win->d_gate, win->d_probe, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_decode_meter_window(igdsp_ctx *ctx, uint32_t layout, const uint8_t *d_packets, const uint16_t *d_sizes, const uint8_t *d_codec,
                              const uint8_t *d_radio, uint32_t C, uint32_t F, uint32_t pkt_stride, uint32_t hdr_bytes,
                              igdsp_frame_stats *d_stats, igdsp_rtp_info *d_info, igdsp_aggregate *d_agg, uint32_t rank,
                              const igdsp_window *win, void *stream)
{
    if (!ctx || layout > IGDSP_PKT_MIXED) return IGDSP_EINVAL;
    if (int rc = check_window(ctx, win)) return rc;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_packets || !d_codec || rank >= IGDSP_AGG_MAX_RANKS) return IGDSP_EINVAL;
    if (layout == IGDSP_PKT_MIXED && !d_radio) return IGDSP_EINVAL;
    const bool fused = C % 64u == 0u && win->d_work != nullptr;
    if (!d_stats && !fused) return fail(ctx, IGDSP_EINVAL, "decode_meter_window: d_stats may only be NULL on the fused path (n_channels % 64 == 0, d_work given)");
    if (int rc = check_shape(C, F, IGDSP_SAMPLES_PER_FRAME)) return rc;
    // the argument rules of the three fused entries
    uint32_t stride = 0, hdr = 20;
    const uint8_t *radio = nullptr;
    const uint16_t *sizes = nullptr;
    if (layout == IGDSP_PKT_SLOTS) {
        if (reinterpret_cast<uintptr_t>(d_packets) & 15u) return fail(ctx, IGDSP_EINVAL, "decode_meter_window: slots need 16-byte alignment");
    } else {
        stride = pkt_stride; sizes = d_sizes;
        if (layout == IGDSP_PKT_PACKED) {
            if (hdr_bytes != 12u && hdr_bytes != 20u) return IGDSP_EINVAL;
            hdr = hdr_bytes;
        } else { hdr = 12; radio = d_radio; }
        const uint32_t need = (layout == IGDSP_PKT_MIXED ? 20u : hdr) + IGDSP_SAMPLES_PER_FRAME;
        if (pkt_stride < need || pkt_stride < 20u || (pkt_stride & 3u) || pkt_stride > 2048u || (uint64_t)C * F * pkt_stride > 0xFFFFFFFFull * 4ull ||
            (reinterpret_cast<uintptr_t>(d_packets) & 3u) || (reinterpret_cast<uintptr_t>(d_sizes) & 1u))
            return IGDSP_EINVAL;
    }
    if (((uint64_t)C * F) % 64u || (reinterpret_cast<uintptr_t>(d_stats) & 15u) || (reinterpret_cast<uintptr_t>(d_info) & 7u))
        return fail(ctx, IGDSP_EINVAL, "decode_meter_window needs C*F % 64 == 0, 16-byte aligned stats, 8-byte aligned info");
    const uint32_t alarm = win->probe_alarm ? win->probe_alarm : IGDSP_PROBE_ALARM;
    hipStream_t s = pick(ctx, stream);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (fused) {
        // channel-group-major fused kernel: the windows live in registers; at least one unit per resident wave, a segment is
        // never shorter than 8 frames nor longer than 65 535 (silent / clipped counts of a unit are 16 bits)
        igdsp::WinArgs w;
        w.work = static_cast<uint4 *>(win->d_work);
        w.gate_mask = ((win->gate_mode & IGDSP_GATE_SQU) ? 0x10000000u : 0u) | ((win->gate_mode & IGDSP_GATE_PTT) ? 0xe0000000u : 0u);   // Functions.cpp:1160, 1136
        w.alarm = alarm; w.n_groups = C / 64u; w.F = F;
        // (16 units per CU for its 12 waves: at 65 536 channels 4 segments — a third of the waves take a second unit — measured
        // 0.2873-0.2929 ms against 0.2973-0.3023 with 3 segments = one unit per wave, 0.2906-0.2954 with 5, 0.2903-0.2957 with 8)
        const uint32_t want = (uint32_t)ctx->cus * 16u;
        uint32_t n_seg = w.n_groups >= want ? 1u : (want + w.n_groups - 1u) / w.n_groups;
        if (const char *e = std::getenv("IGDSP_WIN_NSEG")) n_seg = (uint32_t)std::max(1, std::atoi(e));   // experiments
        n_seg = std::max(1u, std::min(std::min(n_seg, kWinMaxSeg), std::max(1u, F / 8u)));
        if (F / n_seg > 65535u) return fail(ctx, IGDSP_ERANGE, "decode_meter_window: more than 8 x 65535 frames per launch");
        w.n_seg = n_seg;
        // Block-owned form (the default where it fits): a block owns gpb = 4, 2 or 1 consecutive channel groups for the launch,
        // hands their items to its waves in (frame, group) order and keeps their windows and runs in its LDS — the item-level
        // balance of the time-major kernels inside a block, no summaries, no finish kernel.  gpb: as many groups per block as
        // still give every CU a block; taken up to one round of blocks and when the blocks fill whole rounds of the CUs to 85 % (a launch
        // of 1.25 rounds would idle 3/8 of the chip in its second round: the register form above has no such steps).  The packed LDS counters hold
        // 255 frames: longer launches go out as equal parts on the stream (hold / probe / the aggregate carry across them).
        {
            const uint32_t cus = (uint32_t)std::max(1, ctx->cus);
            uint32_t gpb = 1u;
            for (uint32_t g = 4u; g > 1u; g >>= 1) if (w.n_groups % g == 0u && w.n_groups / g >= cus) { gpb = g; break; }
            if (const char *e = std::getenv("IGDSP_WIN_GPB")) { const uint32_t g = (uint32_t)std::atoi(e); if ((g == 1u || g == 2u || g == 4u) && w.n_groups % g == 0u) gpb = g; }   // tests
            const uint32_t blocks = w.n_groups / gpb, rounds = (blocks + cus - 1u) / cus;
            // (measured around the tuned 65 536 channels, register form / block form ms: 2 048 ch 0.0992 / 0.0658, 8 192 ch 0.1014 / 0.0685, 12 288 ch
            // 0.1021 / 0.0705 — up to one round of blocks the block form always wins, the register form walks its segments serially — 24 576 ch
            // 0.1218 / 0.1397, 49 152 ch 0.2332 / 0.2559: 1.5 rounds idle half the chip in the second)
            bool blk = rounds == 1u || (uint64_t)blocks * 100u >= (uint64_t)rounds * cus * 85u;
            if (const char *e = std::getenv("IGDSP_WIN_BLK")) blk = std::atoi(e) != 0;       // experiments and tests: 0 = never, 1 = always
            if (blk) {
                w.gpb = gpb; w.gsh = gpb == 4u ? 2u : (gpb == 2u ? 1u : 0u);
                w.hold = win->d_hold; w.gate = win->d_gate; w.probe = win->d_probe;
                const uint32_t parts = (F + 254u) / 255u;
                const uint64_t pkt_frame = (uint64_t)C * (stride ? stride : (uint32_t)IGDSP_SLOT_BYTES);
                for (uint32_t k = 0; k < parts; ++k) {
                    const uint32_t f0 = (uint32_t)(((uint64_t)F * k) / parts), f1 = (uint32_t)(((uint64_t)F * (k + 1u)) / parts);
                    const uint64_t r0 = (uint64_t)f0 * C;
                    w.F = f1 - f0;
                    HIP_TRY(ctx, launch_decode_meter_rtp(cfg_of(ctx, s), d_packets + (uint64_t)f0 * pkt_frame, sizes ? sizes + r0 : nullptr, d_codec, C, f1 - f0, stride, hdr,
                                                         d_stats ? d_stats + r0 : nullptr, d_info ? d_info + r0 : nullptr, d_agg, rank, s, radio, &w));
                }
                return IGDSP_OK;
            }
        }
        HIP_TRY(ctx, launch_decode_meter_rtp(cfg_of(ctx, s), d_packets, sizes, d_codec, C, F, stride, hdr, d_stats, d_info, d_agg, rank, s, radio, &w));
        HIP_TRY(ctx, launch_window_finish(w.work, C, n_seg, alarm, win->d_hold, win->d_gate, win->d_probe, s));
        return IGDSP_OK;
    }
    // other channel counts: the plain fused kernel, then the record-wise window fold on the same stream
    if (!d_info) return fail(ctx, IGDSP_EINVAL, "decode_meter_window: channel counts that are not multiples of 64 (or a window without d_work) need d_info");
    HIP_TRY(ctx, launch_decode_meter_rtp(cfg_of(ctx, s), d_packets, sizes, d_codec, C, F, stride, hdr, d_stats, d_info, d_agg, rank, s, radio));
    HIP_TRY(ctx, launch_window_update(d_stats, d_info, nullptr, C, F, IGDSP_SAMPLES_PER_FRAME, win->gate_mode, alarm, win->d_hold, win->d_gate, win->d_probe, s));
    return IGDSP_OK;
}

int igdsp_wav_expand(igdsp_ctx *ctx, const uint8_t *d_payload, uint32_t C, uint32_t F, uint32_t n, uint32_t rate,
                     uint8_t *d_files, uint64_t file_stride, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    if ((uint64_t)C * F == 0) return IGDSP_OK;
    if (!d_payload || !d_files) return IGDSP_EINVAL;
    if (int rc = check_shape(C, F, n)) return rc;
    const uint64_t file_bytes = 44ull + 2ull * F * n;
    if (file_stride < file_bytes || 2ull * F * n > 0xFFFFFFFFull - 36ull) return IGDSP_EINVAL;   // the header's sizes are 32-bit
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_wav_expand(cfg_of(ctx, pick(ctx, stream)), d_payload, C, F, n, rate, d_files, file_stride, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_g726_reorder(igdsp_ctx *ctx, const uint8_t *d_in, uint8_t *d_out, uint64_t n_bytes, int mode, void *stream)
{
    if (!ctx || mode < 1 || mode > 4) return IGDSP_EINVAL;
    if (n_bytes == 0) return IGDSP_OK;
    if (!d_in || !d_out) return IGDSP_EINVAL;
    const uint64_t group = (mode == 2) ? 3 : (mode == 4 ? 5 : 1);
    if (n_bytes % group) return IGDSP_EINVAL;               // the reference over-reads on partial groups
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_g726(cfg_of(ctx, pick(ctx, stream)), d_in, d_out, n_bytes, mode, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_gen_uniform(igdsp_ctx *ctx, uint8_t *d_out, uint64_t n_bytes, uint64_t seed, uint64_t first_byte, void *stream)
{
    if (!ctx || (!d_out && n_bytes)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_gen_uniform(d_out, n_bytes, seed, first_byte, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_stream_read(igdsp_ctx *ctx, const void *d_src, size_t bytes, uint64_t *d_sink, void *stream)
{
    if (!ctx || !d_src || !d_sink || (reinterpret_cast<uintptr_t>(d_src) & 15u)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_stream_read(cfg_of(ctx, pick(ctx, stream)), d_src, bytes, d_sink, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_probe_placement(igdsp_ctx *ctx, const void *d_in, size_t bytes, void *d_out, uint32_t reps, float *ms_per_launch, void *stream)
{
    if (!ctx || !d_in || !ms_per_launch || reps == 0 || bytes < 10240u || (reinterpret_cast<uintptr_t>(d_in) & 15u) ||
        (reinterpret_cast<uintptr_t>(d_out) & 15u))
        return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    void *scratch = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (!d_out) {
        if (hipMalloc(&scratch, bytes / 10u + 4096u) != hipSuccess) return fail(ctx, IGDSP_ENOMEM, "probe scratch");
        d_out = scratch;
    }
    hipError_t e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    for (int i = 0; i < 3 && e == hipSuccess; ++i) e = launch_stream_rw(cfg_of(ctx, s), d_in, bytes, d_out, s);
    if (e == hipSuccess) e = hipEventRecord(a, s);
    for (uint32_t i = 0; i < reps && e == hipSuccess; ++i) e = launch_stream_rw(cfg_of(ctx, s), d_in, bytes, d_out, s);
    if (e == hipSuccess) e = hipEventRecord(b, s);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (scratch) (void)hipFree(scratch);
    if (e != hipSuccess) return fail(ctx, IGDSP_EDEVICE, "igdsp_probe_placement", e);
    *ms_per_launch = ms / (float)reps;
    return IGDSP_OK;
}

// Measurement / test helper (not in include/igdsp.h): what `n_calls` media threads do between two ticks, in one native loop —
// `frames_per_call` calls of igdsp_on_rtp_frame for each of the calls first_call .. first_call + n_calls - 1, frame f of call k
// taken from payloads[(f * n_calls + k) % n_payloads][payloadlen].  Returns the number of calls that did not return IGDSP_OK.
int igdsp_internal_stage_many(igdsp_ctx *ctx, int32_t first_call, uint32_t n_calls, uint32_t frames_per_call, uint8_t pt,
                              const uint8_t *payloads, uint32_t n_payloads, uint32_t payloadlen)
{
    if (!ctx || !payloads || n_payloads == 0) return IGDSP_EINVAL;
    int bad = 0;
    for (uint32_t f = 0; f < frames_per_call; ++f)
        for (uint32_t k = 0; k < n_calls; ++k)
            if (igdsp_on_rtp_frame(ctx, first_call + (int32_t)k, pt, payloads + (size_t)((f * n_calls + k) % n_payloads) * payloadlen, payloadlen) != IGDSP_OK) ++bad;
    return bad;
}

// Test-only (not in include/igdsp.h): the table-driven compressor the fused round-trip kernel uses, on arbitrary PCM.
int igdsp_internal_encode_table(igdsp_ctx *ctx, const int16_t *d_pcm, const uint8_t *d_codec, uint32_t C, uint32_t F, uint32_t n,
                                uint8_t *d_out, int variant, void *stream)
{
    if (!ctx || !d_pcm || !d_codec || !d_out || (variant != IGDSP_ENC_SUN16 && variant != IGDSP_ENC_G191)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_encode_table(cfg_of(ctx, pick(ctx, stream)), d_pcm, d_codec, C, F, n, d_out, variant, pick(ctx, stream)));
    return IGDSP_OK;
}

// Calibration-only (not in include/igdsp.h): bare load/store kernel with the meter kernel's exact traffic
// (10 KiB read + 1 KiB record store per super-chunk); d_dst needs bytes / 10 bytes.
int igdsp_internal_stream_rw(igdsp_ctx *ctx, const void *d_src, size_t bytes, void *d_dst, void *stream)
{
    if (!ctx || !d_src || !d_dst || (reinterpret_cast<uintptr_t>(d_src) & 15u) || (reinterpret_cast<uintptr_t>(d_dst) & 15u)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_stream_rw(cfg_of(ctx, pick(ctx, stream)), d_src, bytes, d_dst, pick(ctx, stream)));
    return IGDSP_OK;
}

// Calibration-only (not in include/igdsp.h): the meter's 10 : 1 traffic with the record stores of k consecutive super-chunks clustered
int igdsp_internal_stream_cluster(igdsp_ctx *ctx, const void *d_src, size_t bytes, void *d_dst, int k, void *stream)
{
    if (!ctx || !d_src || !d_dst || ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 15u)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_stream_cluster(cfg_of(ctx, pick(ctx, stream)), d_src, bytes, d_dst, k, pick(ctx, stream)));
    return IGDSP_OK;
}

// Calibration-only (not in include/igdsp.h): the dword-aligned piece pattern of the packed packet / strided kernels, no per-sample work
// (launch_stream_pieces).  src needs n_items * 64 * stride + 16 bytes, dst n_items KiB, dst2 (optional) n_items * 512 bytes.
int igdsp_internal_stream_pieces(igdsp_ctx *ctx, const void *d_src, uint32_t n_items, uint32_t stride, uint32_t hdr, int mode, int rows, void *d_dst, void *d_dst2, void *stream)
{
    if (!ctx || !d_src || !d_dst || (stride & 3u) || stride < 16u * (uint32_t)(rows - (mode == 0 ? 2 : 1))) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_stream_pieces(cfg_of(ctx, pick(ctx, stream)), d_src, n_items, stride, hdr, mode, rows, d_dst, d_dst2, pick(ctx, stream)));
    return IGDSP_OK;
}

// Calibration-only (not in include/igdsp.h): the packed-packet piece stream in the channel-group-major order of the fused window kernel
int igdsp_internal_stream_walk(igdsp_ctx *ctx, const void *d_src, uint32_t n_items, uint32_t stride, uint32_t hdr, uint32_t groups, uint32_t n_seg,
                               uint32_t trickle, void *d_dst, void *d_dst2, void *stream)
{
    if (!ctx || !d_src || !d_dst || (stride & 3u) || n_seg == 0 || (groups && n_items % groups)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_stream_walk(cfg_of(ctx, pick(ctx, stream)), d_src, n_items, stride, hdr, groups, n_seg, trickle, d_dst, d_dst2, pick(ctx, stream)));
    return IGDSP_OK;
}

// Calibration-only (not in include/igdsp.h): bare read : write mix, r and w 1 KiB pieces per wave item
// (pairs built: 0:8, 8:8, 8:4, 4:8, 10:1, 10:0, 8:1, 8:2, 20:2, 5:1; `waves` per block 1..16); src needs n_items * r KiB, dst n_items * w KiB.
int igdsp_internal_stream_mix(igdsp_ctx *ctx, const void *d_src, void *d_dst, uint32_t n_items, int r, int w, int waves, void *stream)
{
    // the source may be only dword aligned: that is what the calibration of misaligned 16-byte loads needs
    if (!ctx || !d_src || !d_dst || (reinterpret_cast<uintptr_t>(d_src) & 3u) || (reinterpret_cast<uintptr_t>(d_dst) & 15u)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_stream_mix(cfg_of(ctx, pick(ctx, stream)), d_src, d_dst, n_items, r, w, waves, pick(ctx, stream)));
    return IGDSP_OK;
}

// same, odd items write into a second window (d_dst2 addressed like d_dst) and, if d_src2 is given, read from a second one
int igdsp_internal_stream_mix2(igdsp_ctx *ctx, const void *d_src, void *d_dst, void *d_dst2, uint32_t n_items, int r, int w, int waves, void *stream,
                               const void *d_src2)
{
    if (!ctx || !d_src || !d_dst || !d_dst2 || (reinterpret_cast<uintptr_t>(d_src2) & 15u) || ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(d_dst2)) & 15u)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_stream_mix(cfg_of(ctx, pick(ctx, stream)), d_src, d_dst, n_items, r, w, waves, pick(ctx, stream), d_dst2, d_src2));
    return IGDSP_OK;
}

// Diagnostic-only (not in include/igdsp.h): cycle stamps of the chunk32 kernel, 8 x u64 per wavefront
// {t_begin, t_lut_ready, t_end, sum load-wait, sum process, iterations, sum frame-reduce, xcc id}.
int igdsp_internal_diag_chunk32(igdsp_ctx *ctx, const uint8_t *d_payload, const uint8_t *d_codec, uint32_t C, uint32_t F,
                                igdsp_frame_stats *d_stats, uint64_t *d_diag, void *stream)
{
    if (!ctx || !d_payload || !d_codec || !d_stats || !d_diag || C < 32) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_diag_chunk32(cfg_of(ctx, pick(ctx, stream)), d_payload, d_codec, C, F, d_stats, d_diag, pick(ctx, stream)));
    return IGDSP_OK;
}

// ---------------------------------------------------------------- memory helpers
int igdsp_dev_alloc(igdsp_ctx *ctx, void **d_ptr, size_t bytes)
{
    if (!ctx || !d_ptr) return IGDSP_EINVAL;
    *d_ptr = nullptr;
    if (bytes == 0) return IGDSP_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return IGDSP_ENODEV;
    hipError_t e = hipMalloc(d_ptr, bytes);
    return e == hipSuccess ? IGDSP_OK : fail(ctx, IGDSP_ENOMEM, "hipMalloc", e);
}

int igdsp_dev_free(igdsp_ctx *ctx, void *d_ptr)
{
    if (!ctx) return IGDSP_EINVAL;
    if (!d_ptr) return IGDSP_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipFree(d_ptr));
    return IGDSP_OK;
}

int igdsp_dev_alloc_far(igdsp_ctx *ctx, void **d_ptr, size_t bytes, const void *d_in, size_t in_bytes, uint32_t max_tries,
                        size_t spacer_bytes, float *ms_first, float *ms_kept)
{
    if (!ctx || !d_ptr || !d_in || bytes == 0 || in_bytes < 10240u || max_tries == 0) return IGDSP_EINVAL;
    *d_ptr = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return IGDSP_ENODEV;
    if (spacer_bytes == 0) spacer_bytes = (size_t)12 << 30;
    const size_t cand_bytes = std::max(bytes, in_bytes / 10u + 4096u);       // the probe writes in_bytes / 10
    std::vector<void *> spacers;
    void *best = nullptr;
    float t_best = 0.f, t_first = 0.f;
    int rc = IGDSP_OK;
    for (uint32_t k = 0; k < max_tries; ++k) {
        if (k > 0) {
            void *sp = nullptr;
            if (hipMalloc(&sp, spacer_bytes) != hipSuccess) { (void)hipGetLastError(); break; }   // out of memory: stop widening
            spacers.push_back(sp);
        }
        void *cand = nullptr;
        if (hipMalloc(&cand, cand_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        float ms = 0.f;
        rc = igdsp_probe_placement(ctx, d_in, in_bytes, cand, 6, &ms, nullptr);
        if (rc != IGDSP_OK) { (void)hipFree(cand); break; }
        if (k == 0) t_first = ms;
        if (best == nullptr || ms < t_best) {
            if (best) (void)hipFree(best);
            best = cand; t_best = ms;
        } else {
            (void)hipFree(cand);
        }
        if (t_best < 0.92f * t_first) break;                                 // another class found
    }
    for (void *sp : spacers) (void)hipFree(sp);
    if (rc != IGDSP_OK) { if (best) (void)hipFree(best); return rc; }
    if (!best) return fail(ctx, IGDSP_ENOMEM, "igdsp_dev_alloc_far");
    *d_ptr = best;
    if (ms_first) *ms_first = t_first;
    if (ms_kept) *ms_kept = t_best;
    return IGDSP_OK;
}

int igdsp_copy_h2d(igdsp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !h_src))) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return IGDSP_OK;
}

int igdsp_copy_d2h(igdsp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (!ctx || (bytes && (!h_dst || !d_src))) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return IGDSP_OK;
}

int igdsp_dev_memset(igdsp_ctx *ctx, void *d_ptr, int value, size_t bytes)
{
    if (!ctx || (bytes && !d_ptr)) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemset(d_ptr, value, bytes));
    return IGDSP_OK;
}

int igdsp_sync(igdsp_ctx *ctx, void *stream)
{
    if (!ctx) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t pos;
    const uint32_t launches = queue_mark(ctx, pick(ctx, stream), &pos);
    HIP_TRY(ctx, hipStreamSynchronize(pick(ctx, stream)));
    queue_release_if_idle(ctx, pick(ctx, stream), pos, launches);      // an idle stream gives its work-counter pair back
    return IGDSP_OK;
}

// ---------------------------------------------------------------- timers (HIP events on the launch stream)
struct igdsp_timer { hipEvent_t a, b; };

int igdsp_timer_create(igdsp_ctx *ctx, void **timer)
{
    if (!ctx || !timer) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    igdsp_timer *t = new (std::nothrow) igdsp_timer();
    if (!t) return IGDSP_ENOMEM;
    if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->b) != hipSuccess) { delete t; return fail(ctx, IGDSP_EDEVICE, "hipEventCreate"); }
    *timer = t;
    return IGDSP_OK;
}

int igdsp_timer_destroy(igdsp_ctx *ctx, void *timer)
{
    if (!ctx || !timer) return IGDSP_EINVAL;
    igdsp_timer *t = (igdsp_timer *)timer;
    (void)hipEventDestroy(t->a); (void)hipEventDestroy(t->b);
    delete t;
    return IGDSP_OK;
}

int igdsp_timer_start(igdsp_ctx *ctx, void *timer, void *stream)
{
    if (!ctx || !timer) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipEventRecord(((igdsp_timer *)timer)->a, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_timer_stop(igdsp_ctx *ctx, void *timer, void *stream)
{
    if (!ctx || !timer) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipEventRecord(((igdsp_timer *)timer)->b, pick(ctx, stream)));
    return IGDSP_OK;
}

int igdsp_timer_elapsed_ms(igdsp_ctx *ctx, void *timer, float *ms)
{
    if (!ctx || !timer || !ms) return IGDSP_EINVAL;
    igdsp_timer *t = (igdsp_timer *)timer;
    HIP_TRY(ctx, hipEventSynchronize(t->b));
    HIP_TRY(ctx, hipEventElapsedTime(ms, t->a, t->b));
    return IGDSP_OK;
}

}  // extern "C"
