// igdsp_io.hip — placement-aware allocation of the hot path's input / output buffers (igdsp_io_alloc, include/igdsp.h).
//
// Why it exists.  On MI355X a kernel that READS one class of device memory and WRITES another runs ~13 % faster than one
// that reads and writes the same class, and a bulk write stream spread over two classes (neither the inputs') is another
// 5-8 % faster (three classes of ~96 GB, in runs of tens of GiB of consecutive allocations — consistent with
// the three stack-ID ranks of the 12-high HBM3E stacks: write-to-read turnaround is paid inside a rank).  Which class an
// allocation lands in is not visible through any API and differs per process, so the only way to place buffers is to
// measure.  Round 1 did that in bench.py (a 200 GB arena and timed launches); a host that followed INTEGRATION.md got the
// slow case.  This file moves it into the product:
//
//   * physical memory is taken in CHUNKS (hipMemCreate, 128 MiB) and each chunk is classified by timing the bare
//     read + record-store stream (the meter kernel's traffic, k_stream_rw) reading the caller's INPUT buffers and writing
//     the chunk;
//   * buffers are virtual address ranges (hipMemAddressReserve) onto which chunks of the wanted class are mapped:
//     INPUT buffers first (their class is "A" by definition), RECORD buffers from chunks of another class, BULK buffers
//     with their first half from one non-A class and their second half from the other (the queue-driven kernels visit the
//     two halves of a bulk output alternately, spread_batch in igdsp_device.h);
//   * exploration is sparse (every 16th chunk is probed until a new class shows up, then its neighbours) and bounded by
//     `explore_limit_bytes`; everything not mapped is released before returning.
//
// When the virtual-memory API is unavailable, the inputs are too small for the probe to mean anything (< 512 MiB: the
// batch lives in the 256 MiB Infinity Cache anyway) or no second class is found inside the limit, the buffers are still
// returned — consecutive chunks / plain hipMalloc — and the report says so (placed = 0).  No CPU path is involved anywhere.
#include "igdsp_ctx.h"

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace igdsp;

struct igdsp_io_set {
    // cls[i] = class label of handles[i] in the context's labelling (igdsp_ctx::io_spare; 255 = unknown), valid while epoch matches
    struct Map { void *va = nullptr; size_t bytes = 0; std::vector<hipMemGenericAllocationHandle_t> handles; std::vector<uint8_t> cls; };
    uint32_t epoch = 0;
    std::vector<Map> maps;            // VMM path: one reserved range per buffer, chunk handles mapped back to back
    std::vector<void *> plain;        // fallback path: hipMalloc'ed buffers
    size_t chunk = 0;
    int device = 0;
};

namespace {

constexpr size_t kSrcChunks = 10;     // chunks behind a probe source: 1.25 GiB streamed per probe launch, far more than the 256 MiB Infinity Cache

struct Chunk {
    hipMemGenericAllocationHandle_t h{};
    float tA = -1.f;                  // probe time as the WRITE side against source A (< 0: not timed)
    float tB = -1.f;                  // ... against source B
    bool used = false;                // handed to a buffer (not to be released)
    bool mapped = false;              // currently mapped on its scratch slot
    bool in_src = false;              // currently part of a probe source
    int cls = -1;                     // class label once established (igdsp_ctx::io_spare), -1 unknown
};

// A probe source: kSrcChunks chunks mapped back to back on an address range of their own.  A range is mapped ONCE: on this
// stack an address that was un-mapped and mapped again keeps reaching the OLD chunk (igdsp_internal_vmm_remap_check), so a new
// source always gets a new range.
struct Source { void *va = nullptr; size_t n = 0; std::vector<size_t> idx; };

struct Explorer {
    igdsp_ctx *ctx = nullptr;
    hipStream_t s = nullptr;
    size_t chunk = 0;
    size_t va_align = 0;              // alignment asked of every address reservation
    hipMemAllocationProp prop{};
    hipMemAccessDesc acc{};
    void *cand_va = nullptr;          // scratch address space for probing: slot idx belongs to chunk idx, mapped at most once
    size_t cand_bytes = 0;
    bool debug = false;
    hipEvent_t ea = nullptr, eb = nullptr;
    std::vector<Chunk> chunks;
    std::vector<Source> sources;
    size_t limit_chunks = 0;
    size_t probe_n = 0;
    uint32_t probes = 0;

    bool ensure(size_t idx)
    {
        while (chunks.size() <= idx) {
            if (chunks.size() >= limit_chunks) return false;
            Chunk c;
            if (hipMemCreate(&c.h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
            chunks.push_back(c);
        }
        return true;
    }
    void unmap_scratch(size_t idx)
    {
        if (chunks[idx].mapped) { (void)hipMemUnmap((char *)cand_va + idx * chunk, chunk); chunks[idx].mapped = false; }
    }
    // time the bare read(src) + record-store(dst) stream: 2 untimed + 4 timed launches
    bool time_pair(const void *rd, void *dst, float *ms)
    {
        hipError_t e = hipSuccess;
        for (int i = 0; i < 2 && e == hipSuccess; ++i) e = launch_stream_rw(cfg_of(ctx, s), rd, probe_n, dst, s);
        if (e == hipSuccess) e = hipEventRecord(ea, s);
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = launch_stream_rw(cfg_of(ctx, s), rd, probe_n, dst, s);
        if (e == hipSuccess) e = hipEventRecord(eb, s);
        if (e == hipSuccess) e = hipEventSynchronize(eb);
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, ea, eb);
        if (e != hipSuccess) { (void)hipGetLastError(); return false; }
        *ms = t / 4.f;
        ++probes;
        return true;
    }
    // time chunk idx as the write side of (source -> chunk); the chunk is mapped on its own scratch slot for that
    bool probe(size_t idx, const Source &src, float *ms, const char *tag)
    {
        char *at = (char *)cand_va + idx * chunk;
        Chunk &c = chunks[idx];
        if (c.in_src) return false;
        if (!c.mapped) {
            if (hipMemMap(at, chunk, 0, c.h, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
            c.mapped = true;
            if (hipMemSetAccess(at, chunk, &acc, 1) != hipSuccess) { (void)hipGetLastError(); return false; }
        }
        const bool ok = time_pair(src.va, at, ms);
        if (debug && ok) std::fprintf(stderr, "[igdsp_io] chunk %zu vs %s: %.4f ms\n", idx, tag, *ms);
        return ok;
    }
    // a new probe source from kSrcChunks chunks (each un-mapped from its scratch slot first: one mapping per chunk at a time)
    bool make_source(const std::vector<size_t> &idx, size_t *which)
    {
        Source S;
        S.n = idx.size();
        if (hipMemAddressReserve(&S.va, S.n * chunk, va_align, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        sources.push_back(S);
        Source &T = sources.back();
        for (size_t k = 0; k < idx.size(); ++k) {
            unmap_scratch(idx[k]);
            if (hipMemMap((char *)T.va + k * chunk, chunk, 0, chunks[idx[k]].h, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
            T.idx.push_back(idx[k]);
            chunks[idx[k]].in_src = true;
            if (hipMemSetAccess((char *)T.va + k * chunk, chunk, &acc, 1) != hipSuccess) { (void)hipGetLastError(); return false; }
        }
        *which = sources.size() - 1;
        return true;
    }
    void drop_source(size_t which)
    {
        Source &S = sources[which];
        for (size_t k = 0; k < S.idx.size(); ++k) { (void)hipMemUnmap((char *)S.va + k * chunk, chunk); chunks[S.idx[k]].in_src = false; }
        S.idx.clear();                 // the address range stays reserved until cleanup and is never mapped again
    }
    void cleanup()
    {
        for (size_t i = 0; i < chunks.size(); ++i) unmap_scratch(i);
        for (size_t w = 0; w < sources.size(); ++w) drop_source(w);
        sources.clear();
        for (auto &c : chunks) if (!c.used) (void)hipMemRelease(c.h);
        chunks.clear();
        if (ea) (void)hipEventDestroy(ea);
        if (eb) (void)hipEventDestroy(eb);
        ea = eb = nullptr; cand_va = nullptr;     // address ranges are never returned: see igdsp_io_free
        (void)hipGetLastError();
    }
};

bool map_chunks(igdsp_io_set::Map &m, size_t chunk, const std::vector<hipMemGenericAllocationHandle_t> &hs, size_t first_slot,
                const hipMemAccessDesc &acc)
{
    for (size_t i = 0; i < hs.size(); ++i) {
        char *at = (char *)m.va + (first_slot + i) * chunk;
        if (hipMemMap(at, chunk, 0, hs[i], 0) != hipSuccess) return false;
        if (hipMemSetAccess(at, chunk, &acc, 1) != hipSuccess) return false;
    }
    return true;
}

}  // namespace

extern "C" {

int igdsp_io_free(igdsp_ctx *ctx, igdsp_io_set *set)
{
    if (!ctx) return IGDSP_EINVAL;
    if (!set) return IGDSP_OK;
    (void)hipSetDevice(set->device);
    (void)hipDeviceSynchronize();
    // Chunks are un-mapped and released; the ADDRESS RANGES stay reserved for the life of the process.  On this stack (ROCm 7.2,
    // gfx950) an address that has been un-mapped — even freed and reserved again — and is then mapped onto another chunk keeps
    // reaching the old one (igdsp_internal_vmm_remap_check; tools/io_place.py prints it), so no address is ever handed back for
    // re-use.  Address space is the only thing this costs (<= ~0.2 TiB of 128 TiB per igdsp_io_alloc call).
    {
        std::lock_guard<std::mutex> g(ctx->io_mu);
        for (auto &m : set->maps)
            for (size_t k = 0; k < ctx->spread_ranges.size();)
                if (ctx->spread_ranges[k].base == (const char *)m.va) ctx->spread_ranges.erase(ctx->spread_ranges.begin() + (long)k); else ++k;
    }
    for (auto &m : set->maps) {
        if (!m.va) continue;
        for (size_t i = 0; i < m.handles.size(); ++i) (void)hipMemUnmap((char *)m.va + i * set->chunk, set->chunk);
        // a chunk whose class is known (and still labelled in the context's current terms) becomes a spare: the next
        // igdsp_io_alloc can map it without probing, and nothing is released for the driver to clear
        std::lock_guard<std::mutex> g(ctx->io_mu);
        for (size_t i = 0; i < m.handles.size(); ++i) {
            const uint8_t c = i < m.cls.size() ? m.cls[i] : 255;
            if (c < 4 && set->epoch == ctx->io_epoch && set->chunk == ctx->io_spare_chunk && ctx->io_spare[c].size() < ctx->io_spare_cap) ctx->io_spare[c].push_back(m.handles[i]);
            else (void)hipMemRelease(m.handles[i]);
        }
    }
    for (void *p : set->plain) if (p) (void)hipFree(p);
    (void)hipGetLastError();
    delete set;
    return IGDSP_OK;
}

}  // extern "C"

void igdsp_io_drop_spares(igdsp_ctx *ctx)
{
    std::lock_guard<std::mutex> g(ctx->io_mu);
    for (auto &v : ctx->io_spare) { for (auto h : v) (void)hipMemRelease(h); v.clear(); }
    (void)hipGetLastError();
}

extern "C" {

// Diagnostic (not in include/igdsp.h): does a device address that was un-mapped and then mapped onto ANOTHER chunk reach
// the new chunk?  Writes 0x11 through address v to chunk X, re-maps v onto chunk Y, writes 0x22 through v, then reads X and Y
// through fresh addresses: bytes[0] / bytes[1] = first byte of X / Y (expected 0x11 / 0x22; X == 0x22 means the second write
// still went to X: a stale translation).  mode 0: hipMemUnmap + hipMemMap on the same reservation; mode 1: the reservation is
// freed (hipMemAddressFree) and reserved again at the same address in between; bytes[2] = 1 when that second reservation did
// come back at the same address.  igdsp_io_alloc maps every address at most once, whatever this reports.
int igdsp_internal_vmm_remap_check(igdsp_ctx *ctx, int mode, int *bytes)
{
    if (!ctx || !bytes) return IGDSP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t gran = 0;
    HIP_TRY(ctx, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t sz = std::max<size_t>(gran, (size_t)64 << 20);
    hipMemGenericAllocationHandle_t X, Y;
    void *v = nullptr, *wx = nullptr, *wy = nullptr;
    HIP_TRY(ctx, hipMemCreate(&X, sz, &prop, 0));
    HIP_TRY(ctx, hipMemCreate(&Y, sz, &prop, 0));
    HIP_TRY(ctx, hipMemAddressReserve(&v, sz, 0, nullptr, 0));
    HIP_TRY(ctx, hipMemAddressReserve(&wx, sz, 0, nullptr, 0));
    HIP_TRY(ctx, hipMemAddressReserve(&wy, sz, 0, nullptr, 0));
    HIP_TRY(ctx, hipMemMap(v, sz, 0, X, 0));
    HIP_TRY(ctx, hipMemSetAccess(v, sz, &acc, 1));
    HIP_TRY(ctx, hipMemsetAsync(v, 0x11, sz, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemUnmap(v, sz));
    bytes[2] = 1;
    if (mode == 1) {
        void *v0 = v;
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipMemAddressFree(v, sz));
        HIP_TRY(ctx, hipMemAddressReserve(&v, sz, 0, v0, 0));
        bytes[2] = v == v0 ? 1 : 0;
    }
    HIP_TRY(ctx, hipMemMap(v, sz, 0, Y, 0));
    HIP_TRY(ctx, hipMemSetAccess(v, sz, &acc, 1));
    HIP_TRY(ctx, hipMemsetAsync(v, 0x22, sz, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemUnmap(v, sz));
    HIP_TRY(ctx, hipMemMap(wx, sz, 0, X, 0));
    HIP_TRY(ctx, hipMemSetAccess(wx, sz, &acc, 1));
    HIP_TRY(ctx, hipMemMap(wy, sz, 0, Y, 0));
    HIP_TRY(ctx, hipMemSetAccess(wy, sz, &acc, 1));
    unsigned char bx = 0, by = 0;
    HIP_TRY(ctx, hipMemcpy(&bx, (char *)wx + sz / 2, 1, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(&by, (char *)wy + sz / 2, 1, hipMemcpyDeviceToHost));
    bytes[0] = bx; bytes[1] = by;
    (void)hipMemUnmap(wx, sz); (void)hipMemUnmap(wy, sz);
    (void)hipMemRelease(X); (void)hipMemRelease(Y);
    // the three reservations are deliberately NOT returned: see igdsp_io_free
    return IGDSP_OK;
}

int igdsp_io_alloc(igdsp_ctx *ctx, igdsp_io_buf *bufs, uint32_t n_bufs, size_t explore_limit_bytes, igdsp_io_set **out_set,
                   igdsp_io_report *rep)
{
    if (!ctx || !bufs || !out_set || n_bufs == 0 || n_bufs > 64) return IGDSP_EINVAL;
    *out_set = nullptr;
    igdsp_io_report R;
    std::memset(&R, 0, sizeof R);
    for (uint32_t i = 0; i < n_bufs; ++i) {
        bufs[i].ptr = nullptr;
        if (bufs[i].bytes == 0 || bufs[i].role > IGDSP_IO_BULK) return IGDSP_EINVAL;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto t_start = std::chrono::steady_clock::now();
    igdsp_io_set *set = new (std::nothrow) igdsp_io_set();
    if (!set) return IGDSP_ENOMEM;
    set->device = ctx->device;
    Explorer X;
    X.ctx = ctx;
    X.s = ctx->stream;

    auto finish = [&](int rc) {
        X.cleanup();
        R.setup_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        if (rep) *rep = R;
        if (rc != IGDSP_OK) { igdsp_io_free(ctx, set); for (uint32_t i = 0; i < n_bufs; ++i) bufs[i].ptr = nullptr; }
        else *out_set = set;
        return rc;
    };
    auto plain_path = [&]() {                       // consecutive hipMallocs in the order given: what a host would write itself
        for (uint32_t i = 0; i < n_bufs; ++i) {
            void *p = nullptr;
            if (hipMalloc(&p, bufs[i].bytes) != hipSuccess) { (void)hipGetLastError(); return finish(fail(ctx, IGDSP_ENOMEM, "igdsp_io_alloc: hipMalloc")); }
            set->plain.push_back(p);
            bufs[i].ptr = p;
        }
        return finish(IGDSP_OK);
    };

    int vmm = 0;
    if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, ctx->device) != hipSuccess) { (void)hipGetLastError(); vmm = 0; }
    if (const char *e = std::getenv("IGDSP_IO_PLAIN")) if (std::atoi(e) != 0) vmm = 0;
    if (!vmm) return plain_path();

    X.prop.type = hipMemAllocationTypePinned;
    X.prop.location.type = hipMemLocationTypeDevice;
    X.prop.location.id = ctx->device;
    X.acc.location = X.prop.location;
    X.acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &X.prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) { (void)hipGetLastError(); return plain_path(); }
    size_t chunk = (size_t)128 << 20;               // >= the 1/10 of a probe read that a probe launch writes, and small next to the class runs (3-40 GiB)
    chunk = (chunk + gran - 1) / gran * gran;
    X.chunk = set->chunk = chunk;
    X.va_align = chunk;                             // chunk-aligned addresses: the page tables can then map a chunk with its largest fragments
    if (const char *e = std::getenv("IGDSP_IO_ALIGN_MIB")) X.va_align = (size_t)std::max(0, std::atoi(e)) << 20;
    R.chunk_bytes = chunk;

    // address ranges + chunk counts per buffer
    set->maps.resize(n_bufs);
    std::vector<size_t> nch(n_bufs);
    size_t in_chunks = 0, in_bytes = 0, rec_chunks = 0, bulk_chunks = 0;
    for (uint32_t i = 0; i < n_bufs; ++i) {
        nch[i] = (bufs[i].bytes + chunk - 1) / chunk;
        auto &m = set->maps[i];
        m.bytes = nch[i] * chunk;
        if (hipMemAddressReserve(&m.va, m.bytes, X.va_align, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); m.va = nullptr; return finish(fail(ctx, IGDSP_ENOMEM, "igdsp_io_alloc: hipMemAddressReserve")); }
        if (bufs[i].role == IGDSP_IO_INPUT) { in_chunks += nch[i]; in_bytes += bufs[i].bytes; }
        else if (bufs[i].role == IGDSP_IO_RECORD) rec_chunks += nch[i];
        else bulk_chunks += nch[i];
    }

    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    // default: 50 % of what is free; 85 % when bulk outputs want a THIRD class (the allocator tends to hand that one out last).
    // That is only the ceiling: the search stops as soon as every buffer has its chunks (typically 15-50 GB explored for a
    // two-class set), and later calls are served from the spares this one leaves behind.
    double dflt = bulk_chunks >= 4 ? 0.85 : 0.5;
    if (const char *e = std::getenv("IGDSP_IO_LIMIT_FRAC")) dflt = std::atof(e);
    size_t limit = explore_limit_bytes ? explore_limit_bytes : (size_t)(dflt * (double)free_b);
    limit = std::min(limit, (size_t)(0.9 * (double)free_b));

    // fresh consecutive chunks for what buffer i still lacks (no class wanted / known)
    auto map_fresh = [&](uint32_t i) {
        auto &m = set->maps[i];
        const size_t have = m.handles.size();
        std::vector<hipMemGenericAllocationHandle_t> hs;
        bool ok = true;
        for (size_t k = have; k < nch[i] && ok; ++k) {
            hipMemGenericAllocationHandle_t h;
            ok = hipMemCreate(&h, chunk, &X.prop, 0) == hipSuccess;
            if (ok) hs.push_back(h);
        }
        ok = ok && map_chunks(m, chunk, hs, have, X.acc);
        m.handles.insert(m.handles.end(), hs.begin(), hs.end());      // owned by the set from here on (released by igdsp_io_free)
        m.cls.resize(m.handles.size(), 255);                          // class unknown
        return ok;
    };

    // Below 512 MiB of inputs a launch works out of the 256 MiB Infinity Cache and placement does not matter.
    // (a set of INPUT buffers only is still worth placing once it is larger than the probe source: all of it lands in ONE class)
    const bool want_place = in_bytes >= ((size_t)512 << 20) && ((rec_chunks + bulk_chunks) > 0 || in_chunks > kSrcChunks) && limit / chunk >= 4 * kSrcChunks;
    if (const char *e = std::getenv("IGDSP_IO_SPARE_CHUNKS")) ctx->io_spare_cap = (size_t)std::max(0, std::atoi(e));
    // Served from what an earlier call learnt?  Spare chunks of known class (left by a search, or returned by igdsp_io_free)
    // cover this set when the inputs fit class 0, the records and the bulk outputs' first halves fit the non-0 spares and the
    // second halves the other non-0 class: map them, probe nothing, release nothing (so there is nothing to wait out either).
    if (want_place) {
        std::lock_guard<std::mutex> g(ctx->io_mu);
        auto &S0 = ctx->io_spare[0], &S1 = ctx->io_spare[1], &S2 = ctx->io_spare[2], &S3 = ctx->io_spare[3];
        const bool want_spread = bulk_chunks >= 4;
        const size_t second = want_spread ? bulk_chunks / 2 : 0, first = rec_chunks + bulk_chunks - second;
        if (ctx->io_spare_chunk == chunk && S0.size() >= in_chunks && S1.size() + S2.size() >= first && S3.size() >= second && (!want_spread || S2.size() + S1.size() >= first)) {
            bool okf = true;
            auto take = [&](std::vector<hipMemGenericAllocationHandle_t> &from, uint8_t label, igdsp_io_set::Map &m) {
                m.handles.push_back(from.back()); m.cls.push_back(label); from.pop_back();
            };
            for (uint32_t role = 0; role < 3; ++role)
                for (uint32_t i = 0; i < n_bufs; ++i) {
                    if (bufs[i].role != role) continue;
                    auto &m = set->maps[i];
                    const size_t h2 = (role == IGDSP_IO_BULK && want_spread) ? nch[i] / 2 : 0;
                    for (size_t k = 0; k < nch[i]; ++k) {
                        if (role == IGDSP_IO_INPUT) take(S0, 0, m);
                        else if (k >= nch[i] - h2 && !S3.empty()) take(S3, 3, m);
                        else if (want_spread ? !S2.empty() : !S1.empty()) { if (want_spread) take(S2, 2, m); else take(S1, 1, m); }
                        else if (!S2.empty()) take(S2, 2, m);
                        else if (!S1.empty()) take(S1, 1, m);
                        else take(S3, 3, m);
                    }
                    okf = okf && map_chunks(m, chunk, m.handles, 0, X.acc);
                }
            if (!okf) { (void)hipGetLastError(); return finish(fail(ctx, IGDSP_ENOMEM, "igdsp_io_alloc: mapping spare chunks")); }
            set->epoch = ctx->io_epoch;
            bool spread_ok = want_spread;
            for (uint32_t i = 0; i < n_bufs && spread_ok; ++i)
                if (bufs[i].role == IGDSP_IO_BULK) spread_ok = !set->maps[i].cls.empty() && set->maps[i].cls.front() != set->maps[i].cls.back();
            R.placed = 1; R.bulk_spread = (spread_ok && bulk_chunks > 0) ? 1u : 0u; R.classes_found = R.bulk_spread ? 3u : 2u;
            for (uint32_t i = 0; i < n_bufs; ++i) bufs[i].ptr = set->maps[i].va;
            if (R.bulk_spread)
                for (uint32_t i = 0; i < n_bufs; ++i)
                    if (bufs[i].role == IGDSP_IO_BULK) ctx->spread_ranges.push_back({(const char *)set->maps[i].va, set->maps[i].bytes});
            R.setup_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
            if (rep) *rep = R;
            *out_set = set;
            return IGDSP_OK;
        }
    }
    std::vector<size_t> poolA, poolB, poolC;        // chunk indices: class of the inputs / first other class / second other class
    bool ok = true;
    float lab_fast = 0.f, lab_slow = 0.f, lab_thrB = 0.f, lab_thrC = 0.f;     // thresholds the search ended with (labels of the leftovers)
    bool lab_split = false;
    if (want_place) {
        // a full search re-establishes the class labels: spares and older buffer sets were labelled relative to ANOTHER search's
        // inputs, which may have landed in a different class than this one's
        igdsp_io_drop_spares(ctx);
        { std::lock_guard<std::mutex> g(ctx->io_mu); ctx->io_epoch += 1; ctx->io_spare_chunk = chunk; }
        set->epoch = ctx->io_epoch;
        X.probe_n = kSrcChunks * (chunk - 4096) / 10240 * 10240;       // a probe launch reads this much and writes 1/10 of it into the chunk under test
        X.limit_chunks = limit / chunk;
        X.cand_bytes = X.limit_chunks * chunk;
        X.debug = std::getenv("IGDSP_IO_DEBUG") != nullptr;
        size_t stride = 16;                         // sparse survey: one probe per 2 GiB (classes come in runs of 3-40 GiB of consecutive chunks)
        if (const char *e = std::getenv("IGDSP_IO_STRIDE")) stride = std::max(1, std::atoi(e));
        ok = hipMemAddressReserve(&X.cand_va, X.cand_bytes, X.va_align, nullptr, 0) == hipSuccess;
        ok = ok && hipEventCreate(&X.ea) == hipSuccess && hipEventCreate(&X.eb) == hipSuccess;

        // measured on MI355X: same-class 0.252 ms, other-class 0.219 ms per probe launch (ratio 1.15), spread inside a level < 1 %
        const float kBimodal = 1.08f, kPure = 1.125f, kNear = 1.035f;
        size_t srcA = 0, srcB = 0;
        float tmin = 1e30f, tmax = 0.f;
        std::vector<size_t> surveyed;

        // source A: ten consecutive chunks; every other chunk is ranked by how fast the stream runs when it writes there
        auto seed = [&](size_t first) {
            std::vector<size_t> idx;
            for (size_t k = 0; k < kSrcChunks; ++k) { if (!X.ensure(first + k)) return false; idx.push_back(first + k); }
            return X.make_source(idx, &srcA);
        };
        auto timeA = [&](size_t idx) {               // time chunk idx against source A (once)
            Chunk &c = X.chunks[idx];
            if (c.tA >= 0.f || c.in_src) return true;
            float t = 0.f;
            if (!X.probe(idx, X.sources[srcA], &t, "A")) { ok = false; return false; }
            c.tA = t;
            return true;
        };
        auto survey = [&](bool rescan) {             // every stride-th chunk until two levels are visible and three samples sit on the fast one
            tmin = 1e30f; tmax = 0.f;
            std::vector<size_t> todo = rescan ? surveyed : std::vector<size_t>();
            surveyed.clear();
            size_t next = 0, seen = 0;
            for (;;) {
                size_t idx;
                if (seen < todo.size()) idx = todo[seen];
                else { idx = next; if (!X.ensure(idx)) break; }
                next = std::max(next, idx) + stride;
                ++seen;
                if (X.chunks[idx].in_src) continue;
                if (!timeA(idx)) break;
                surveyed.push_back(idx);
                tmin = std::min(tmin, X.chunks[idx].tA); tmax = std::max(tmax, X.chunks[idx].tA);
                size_t n_fast = 0;
                for (size_t k : surveyed) if (X.chunks[k].tA < kNear * tmin) ++n_fast;
                if (seen >= todo.size() && tmax > kBimodal * tmin && n_fast >= 3 && surveyed.size() >= 8) break;
            }
        };
        ok = ok && seed(0);
        {   // clocks: ~40 ms of the probe stream before anything is compared (the first launches after idle run ~6 % slow)
            float t = 0.f;
            if (ok && X.ensure(kSrcChunks)) { for (int k = 0; k < 28 && ok; ++k) ok = X.probe(kSrcChunks, X.sources[srcA], &t, "warm"); }
        }
        if (ok) survey(false);
        for (int attempt = 0; attempt < 2 && ok && tmax > 1.02f * tmin && tmax < kPure * tmin; ++attempt) {
            // The levels are closer than two pure classes give: the ten consecutive chunks of source A mix classes (a run boundary,
            // or memory that is interleaved chunk by chunk).  The SLOWEST destinations are pure chunks of the class the mixed source
            // holds most of: collect ten of them — the slowest survey sample and its neighbours, then the next slowest — and make
            // them the source.  Everything is timed again against it.
            std::vector<size_t> order(surveyed);
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return X.chunks[a].tA > X.chunks[b].tA; });
            std::vector<size_t> pick;
            const float near_slow = tmax / 1.015f;
            for (size_t si = 0; si < order.size() && pick.size() < kSrcChunks && ok; ++si) {
                const size_t c0 = order[si];
                if (X.chunks[c0].tA < near_slow) break;
                const size_t lo = c0 >= stride ? c0 - stride + 1 : 0;
                for (size_t k = lo; k < c0 + stride && pick.size() < kSrcChunks && ok; ++k) {
                    if (!X.ensure(k) || X.chunks[k].in_src) continue;
                    if (!timeA(k)) break;
                    if (X.chunks[k].tA >= near_slow && std::find(pick.begin(), pick.end(), k) == pick.end()) pick.push_back(k);
                }
            }
            if (pick.size() < kSrcChunks) break;
            if (X.debug) std::fprintf(stderr, "[igdsp_io] levels %.4f / %.4f: source A is mixed, re-seeding from the %zu slowest chunks (first %zu)\n", tmax, tmin, pick.size(), pick[0]);
            X.drop_source(srcA);
            for (auto &c : X.chunks) c.tA = -1.f;
            ok = X.make_source(pick, &srcA);
            if (ok) survey(true);
            ++R.reseeds;
        }
        const bool bimodal = ok && tmax > kBimodal * tmin;
        R.probe_ms_same = tmax;
        R.probe_ms_other = tmin;
        if (bimodal) {
            R.classes_found = 2;
            const float thr_fast = kNear * tmin, thr_slow = tmax / kNear;
            lab_fast = thr_fast; lab_slow = thr_slow;
            auto fast_A = [&](size_t idx) { return !X.chunks[idx].in_src && timeA(idx) && X.chunks[idx].tA < thr_fast; };
            auto slow_A = [&](size_t idx) { return X.chunks[idx].in_src ? false : (timeA(idx) && X.chunks[idx].tA > thr_slow); };
            // Walk the chunk sequence from `idx` and collect `want` chunks that satisfy `pred` (which probes on demand): inside
            // runs that fail, step by `stride` (or to the next chunk that has been timed already); on a hit, walk back over the
            // chunks skipped since the last probe, then go on densely.
            std::vector<char> taken;
            auto walk = [&](size_t idx, size_t want, auto &&pred, std::vector<size_t> &out) {
                auto take = [&](size_t k) { if (taken.size() <= k) taken.resize(k + 1, 0); if (!taken[k]) { taken[k] = 1; out.push_back(k); } };
                while (ok && out.size() < want && X.ensure(idx)) {
                    if (pred(idx)) {
                        size_t lo = idx;
                        while (ok && lo > 0 && X.chunks[lo - 1].tA < 0.f && !X.chunks[lo - 1].in_src && pred(lo - 1)) --lo;
                        for (size_t k = lo; k <= idx && out.size() < want; ++k) take(k);
                        ++idx;
                    } else {
                        size_t nxt = idx + 1;            // skip ahead through un-timed chunks, but stop at one that has been timed already
                        while (nxt < idx + stride && (nxt >= X.chunks.size() || X.chunks[nxt].tA < 0.f)) ++nxt;
                        idx = nxt;
                    }
                }
                std::sort(out.begin(), out.end());
            };
            // inputs: the chunks of source A themselves plus chunks that write slowly against it (the same class)
            if (in_chunks > kSrcChunks) walk(0, in_chunks - kSrcChunks, slow_A, poolA);
            const bool want_spread = bulk_chunks >= 4 && std::getenv("IGDSP_IO_NO_SPREAD") == nullptr;   // (experiments: two classes only)
            // (with a bulk output the pool also has to yield source B and enough members of either class to re-seed it from)
            walk(0, want_spread ? std::max<size_t>(rec_chunks + bulk_chunks + kSrcChunks, 4 * kSrcChunks) : rec_chunks + bulk_chunks, fast_A, poolB);

            // 2nd split, for bulk outputs: which destinations are fast against the inputs' class AND against the first other
            // class?  Source B = the first ten pool chunks (consecutive chunks of one run): a chunk that writes slowly against it
            // shares its class (B), a fast one belongs to the third class (C).  C usually lies tens of GiB further along the
            // allocation sequence, so the walk continues from where the pool ended until the second halves of the bulk buffers
            // are covered, or the exploration limit is reached (then B serves both halves).
            if (ok && want_spread && poolB.size() >= kSrcChunks + rec_chunks + bulk_chunks - bulk_chunks / 2) {
                const size_t need_c = bulk_chunks / 2, need_b = rec_chunks + bulk_chunks - need_c;
                float tbmin = 1e30f, tbmax = 0.f;
                auto timeB = [&](size_t idx) {           // chunk idx against source B (once); false: not a candidate, or a failure
                    Chunk &c = X.chunks[idx];
                    if (c.in_src || !ok) return false;
                    if (c.tB < 0.f) {
                        float t = 0.f;
                        if (!X.probe(idx, X.sources[srcB], &t, "B")) { ok = false; return false; }
                        c.tB = t;
                        tbmin = std::min(tbmin, t); tbmax = std::max(tbmax, t);
                    }
                    return true;
                };
                // Levels of the pool against the current source B, by two-means over the measured times (robust against a stray
                // sample, which max / min are not): cC / cB = centre of the fast / slow group.  A destination is slow against a source
                // in proportion to the share of the source that is of its own class, so with a source that mixes the two classes the
                // two groups are still the two classes, only closer together.
                float cB = 0.f, cC = 0.f, thrB = 0.f, thrC = 0.f;
                size_t nB = 0, nC = 0;
                auto levels = [&]() {
                    float lo = tbmin, hi = tbmax;
                    for (int it = 0; it < 12; ++it) {
                        double sl = 0, sh = 0; size_t nl = 0, nh = 0;
                        for (size_t k : poolB) {
                            const float t = X.chunks[k].tB;
                            if (t < 0.f) continue;
                            if (std::fabs(t - lo) <= std::fabs(t - hi)) { sl += t; ++nl; } else { sh += t; ++nh; }
                        }
                        if (nl) lo = (float)(sl / (double)nl);
                        if (nh) hi = (float)(sh / (double)nh);
                        nC = nl; nB = nh;
                    }
                    cC = lo; cB = hi;
                };
                auto set_thresholds = [&]() { const float mid = 0.5f * (cB + cC), m = 0.2f * (cB - cC); thrB = mid + m; thrC = mid - m; };
                auto is_C = [&](size_t idx) { return fast_A(idx) && timeB(idx) && X.chunks[idx].tB < thrC; };
                auto is_B = [&](size_t idx) { return fast_A(idx) && timeB(idx) && X.chunks[idx].tB > thrB; };
                // Every pool chunk against a candidate source.  0: two well separated levels, or one level at the slow mark (the
                // source is one class and so is the pool).  1: two levels closer than pure classes give (a mixed source: they can
                // interleave chunk by chunk).  2: one level in the middle (the source holds the two classes evenly: separates nothing).
                auto try_source = [&](const std::vector<size_t> &sb) {
                    for (auto &c : X.chunks) c.tB = -1.f;
                    tbmin = 1e30f; tbmax = 0.f;
                    ok = X.make_source(sb, &srcB);
                    for (size_t k : poolB) if (ok) (void)timeB(k);
                    int v = -1;
                    if (ok) {
                        levels();
                        const float sep = cB / cC;
                        if (sep < 1.025f) {
                            const float level = (cB * (float)nB + cC * (float)nC) / (float)std::max<size_t>(1, nB + nC);
                            v = level >= tmax / 1.025f ? 0 : 2;
                            cB = level; cC = level * tmin / tmax;          // all of the pool is class B: the other level is the A test's
                        } else v = sep >= 1.10f ? 0 : 1;
                        set_thresholds();
                    }
                    if (X.debug) std::fprintf(stderr, "[igdsp_io] source B from chunk %zu: levels %.4f (%zu) / %.4f (%zu) -> %s\n", sb[0], cB, nB, cC, nC,
                                              v == 0 ? "one class" : (v == 1 ? "mixed" : (v == 2 ? "evenly mixed" : "failed")));
                    return v;
                };
                int verdict = -1;
                bool have_src = false;
                // candidate sources: ten consecutive pool chunks from the start, every 2nd, every 3rd, then consecutive windows further in
                const size_t picks[][2] = {{0, 1}, {0, 2}, {0, 3}, {kSrcChunks, 1}, {2 * kSrcChunks, 1}, {3 * kSrcChunks, 1}, {1, 2}};
                for (size_t pi = 0; pi < sizeof(picks) / sizeof(picks[0]) && ok && verdict != 0; ++pi) {
                    std::vector<size_t> sb;
                    for (size_t k = picks[pi][0]; k < poolB.size() && sb.size() < kSrcChunks; k += picks[pi][1]) sb.push_back(poolB[k]);
                    if (sb.size() < kSrcChunks) continue;
                    for (int tries = 0; ok; ++tries) {
                        verdict = try_source(sb);
                        have_src = true;
                        if (verdict != 1 || tries == 2) break;
                        // a mixed source: each group it separates is one class — re-seed from ten members of the larger group
                        std::vector<size_t> gb, gc;
                        for (size_t k : poolB) {
                            const Chunk &c = X.chunks[k];
                            if (c.in_src || c.tB < 0.f) continue;
                            if (c.tB > thrB) gb.push_back(k); else if (c.tB < thrC) gc.push_back(k);
                        }
                        std::vector<size_t> &grp = gb.size() >= gc.size() ? gb : gc;
                        if (grp.size() < kSrcChunks) break;
                        grp.resize(kSrcChunks);
                        X.drop_source(srcB);
                        sb = grp;
                        ++R.reseeds;
                    }
                    if (verdict == 1 && cB / cC >= 1.045f) verdict = 0;      // closer than pure classes, yet clearly two groups: good enough to sort by
                    if (verdict != 0 && have_src) { X.drop_source(srcB); have_src = false; }
                }
                if (ok && verdict == 0) {
                    // the pool, classified: source chunks and slow destinations are class B, fast destinations class C, anything between
                    // the levels (a chunk that itself mixes classes) is left out; what is still missing is searched further along
                    std::vector<size_t> cb, cc;
                    for (size_t k : poolB) {
                        const Chunk &c = X.chunks[k];
                        if (c.in_src || c.tB > thrB) cb.push_back(k);
                        else if (c.tB >= 0.f && c.tB < thrC) cc.push_back(k);
                    }
                    taken.assign(taken.size(), 0);
                    for (size_t k : poolB) { if (taken.size() <= k) taken.resize(k + 1, 0); taken[k] = 1; }
                    const size_t from = poolB.back() + 1;
                    if (cc.size() < need_c) walk(from, need_c, is_C, cc);
                    const size_t want_b = need_b + (cc.size() < need_c ? need_c - cc.size() : 0);   // B also serves what C could not
                    if (ok && cb.size() < want_b) walk(from, want_b, is_B, cb);
                    if (X.debug) std::fprintf(stderr, "[igdsp_io] split: class B %zu of %zu chunks, class C %zu of %zu\n", cb.size(), need_b, cc.size(), need_c);
                    // (enough of class C: everything the second halves need, or at least four chunks and half of it.  Requiring four chunks
                    // outright turned down small sets whose two or three C chunks had all been found — 80-byte frames, C 160 GB into the sequence.)
                    if (ok && (cc.size() >= need_c || cc.size() >= std::max<size_t>(4, need_c / 2))) {
                        R.classes_found = 3;
                        poolB = cb;
                        poolC = cc;
                        lab_split = true; lab_thrB = thrB; lab_thrC = thrC;
                    }
                }
                if (have_src) X.drop_source(srcB);
            }
            // Spares for the NEXT set of this size: chunks the sparse survey created but never timed are classified now — about
            // 1.5 ms of probing each, no new memory — until each class has io_spare_cap of them beyond what this set takes; they stay
            // with the context (below) and a later call that they cover is served without a search.
            if (ok && !want_spread && ctx->io_spare_cap > 0) {
                const size_t need0 = (in_chunks > kSrcChunks ? in_chunks - kSrcChunks : 0) + ctx->io_spare_cap, need1 = rec_chunks + bulk_chunks + ctx->io_spare_cap;
                size_t cnt0 = 0, cnt1 = 0;
                for (const auto &c : X.chunks) if (!c.in_src && c.tA >= 0.f) { if (c.tA > thr_slow) ++cnt0; else if (c.tA < thr_fast) ++cnt1; }
                size_t budget = 4 * ctx->io_spare_cap;
                for (size_t k = 0; k < X.chunks.size() && ok && budget > 0 && (cnt0 < need0 || cnt1 < need1); ++k) {
                    Chunk &c = X.chunks[k];
                    if (c.in_src || c.tA >= 0.f) continue;
                    if (!timeA(k)) break;
                    --budget;
                    if (c.tA > thr_slow) ++cnt0; else if (c.tA < thr_fast) ++cnt1;
                }
            }
            // the inputs get source A's own chunks first
            if (ok) {
                std::vector<size_t> a = X.sources[srcA].idx;
                X.drop_source(srcA);
                a.insert(a.end(), poolA.begin(), poolA.end());
                poolA = a;
            }
        }
        R.chunks_explored = (uint32_t)X.chunks.size();
        R.explored_bytes = (uint64_t)X.chunks.size() * chunk;
        R.probes = X.probes;
        (void)hipGetLastError();
    }

    // Hand the picked chunks to their buffers: INPUT buffers from pool A, RECORD buffers and the first halves of the BULK
    // buffers from pool B, the second halves from pool C (from B when there is no third class); a pool that runs dry is topped
    // up from the other output pool, and whatever a buffer still lacks after that comes from fresh consecutive chunks.
    if (ok && (!poolB.empty() || (rec_chunks + bulk_chunks == 0 && !poolA.empty()))) {
        for (size_t i = 0; i < X.chunks.size(); ++i) X.unmap_scratch(i);
        size_t a = 0, b = 0, c = 0;
        bool all = true;
        const bool spread = !poolC.empty();
        std::vector<uint8_t> labels;              // of the chunks the current buffer pulled (igdsp_ctx::io_spare labels)
        auto pull = [&](int pool, size_t n, std::vector<hipMemGenericAllocationHandle_t> &hs) {
            for (size_t k = 0; k < n; ++k) {
                std::vector<size_t> *p = pool == 0 ? &poolA : (pool == 1 ? &poolB : &poolC);
                size_t *cur = pool == 0 ? &a : (pool == 1 ? &b : &c);
                if (*cur >= p->size() && pool != 0) { p = pool == 1 ? &poolC : &poolB; cur = pool == 1 ? &c : &b; }
                if (*cur >= p->size()) return;
                X.chunks[(*p)[*cur]].used = true;
                labels.push_back(p == &poolA ? 0 : (p == &poolC ? 3 : (lab_split ? 2 : 1)));
                hs.push_back(X.chunks[(*p)[(*cur)++]].h);
            }
        };
        for (uint32_t role = 0; role < 3 && ok; ++role)
            for (uint32_t i = 0; i < n_bufs && ok; ++i) {
                if (bufs[i].role != role) continue;
                std::vector<hipMemGenericAllocationHandle_t> hs;
                labels.clear();
                if (role == IGDSP_IO_INPUT) pull(0, nch[i], hs);
                else {
                    const size_t h2 = (role == IGDSP_IO_BULK && spread) ? nch[i] / 2 : 0;  // second-half chunks from class C
                    pull(1, nch[i] - h2, hs);
                    if (hs.size() == nch[i] - h2) pull(2, h2, hs);
                }
                all = all && hs.size() == nch[i];
                auto &m = set->maps[i];
                ok = map_chunks(m, chunk, hs, 0, X.acc);
                m.handles = hs;
                m.cls = labels;
            }
        R.placed = (ok && all) ? 1u : 0u;
        R.bulk_spread = (ok && all && spread && bulk_chunks > 0) ? 1u : 0u;
        if (X.debug && ok) {                        // the finished set against the levels of the search: first INPUT -> every output buffer
            std::fprintf(stderr, "[igdsp_io] pools: A %zu B %zu C %zu chunks; A:", poolA.size(), poolB.size(), poolC.size());
            for (size_t k = 0; k < poolA.size() && k < 12; ++k) std::fprintf(stderr, " %zu", poolA[k]);
            std::fprintf(stderr, "  B:");
            for (size_t k = 0; k < poolB.size() && k < 12; ++k) std::fprintf(stderr, " %zu(%.4f)", poolB[k], X.chunks[poolB[k]].tA);
            std::fprintf(stderr, "\n");
            int in0 = -1;
            for (uint32_t i = 0; i < n_bufs; ++i) if (bufs[i].role == IGDSP_IO_INPUT && (in0 < 0 || nch[i] > nch[in0])) in0 = (int)i;
            for (uint32_t i = 0; i < n_bufs && in0 >= 0 && nch[in0] >= kSrcChunks; ++i) {
                if (bufs[i].role == IGDSP_IO_INPUT) continue;
                for (size_t k = 0; k < nch[i] && k < 3; ++k) {
                    float t = 0.f;
                    if (X.time_pair(set->maps[in0].va, (char *)set->maps[i].va + k * chunk, &t))
                        std::fprintf(stderr, "[igdsp_io] final check: input -> buffer %u chunk %zu: %.4f ms\n", i, k, t);
                }
            }
        }
    }
    // Releasing the exploration chunks leaves the memory system busy for a while: the driver clears freed device memory in the
    // background at ~30-40 GB/s (measured: 104 GB released -> the same buffers stream ~5 % slower and k_meter_chunk64 runs
    // 1.5 % slower for 2-3 s; 56 GB -> 1.5-2 s; 17 GB -> ~0.5 s; then both return to the level of the search).  The call only
    // returns when that is over: at least released bytes / 25 GB/s, and until the finished set streams at the speed it had
    // BEFORE the release (t_ref, taken here; the wait follows the last mapping).  IGDSP_IO_SETTLE=0 skips the wait.
    size_t released_chunks = 0;
    for (const auto &c : X.chunks) if (!c.used) ++released_chunks;
    float t_ref = 0.f;
    const void *settle_rd = nullptr;
    void *settle_wr = nullptr;
    size_t settle_n = 0;
    if (ok && want_place && R.placed) {
        int in0 = -1, out0 = -1;
        for (uint32_t i = 0; i < n_bufs; ++i) {
            if (bufs[i].role == IGDSP_IO_INPUT) { if (in0 < 0 || nch[i] > nch[in0]) in0 = (int)i; }
            else if (out0 < 0) out0 = (int)i;
        }
        if (in0 >= 0 && out0 >= 0) {
            settle_n = std::min(X.probe_n, std::min(nch[in0] * chunk, 10 * (nch[out0] * chunk - 4096)) / 10240 * 10240);
            settle_rd = set->maps[in0].va; settle_wr = set->maps[out0].va;
            const size_t keep = X.probe_n;
            X.probe_n = settle_n;
            float t = 0.f;
            if (settle_n >= ((size_t)256 << 20) && X.time_pair(settle_rd, settle_wr, &t) && X.time_pair(settle_rd, settle_wr, &t)) t_ref = t;
            X.probe_n = keep;
        }
    }
    // Leftovers whose class the search established stay with the context as spares (up to io_spare_cap per class): the next call
    // that they cover maps them without probing.  Pool members first (they were classified with the final thresholds), then any
    // other timed chunk; chunks between two levels, or never timed, are released like before.
    if (ok && want_place && R.placed && lab_fast > 0.f) {
        for (size_t k : poolA) if (!X.chunks[k].used) X.chunks[k].cls = 0;
        for (size_t k : poolB) if (!X.chunks[k].used) X.chunks[k].cls = lab_split ? 2 : 1;
        for (size_t k : poolC) if (!X.chunks[k].used) X.chunks[k].cls = 3;
        std::lock_guard<std::mutex> g(ctx->io_mu);
        for (auto &c : X.chunks) {
            if (c.used || c.in_src) continue;
            if (c.cls < 0 && c.tA >= 0.f) {
                if (c.tA > lab_slow) c.cls = 0;
                else if (c.tA < lab_fast) c.cls = !lab_split ? 1 : (c.tB > lab_thrB ? 2 : ((c.tB >= 0.f && c.tB < lab_thrC) ? 3 : -1));
            }
            if (c.cls >= 0 && ctx->io_spare[c.cls].size() < ctx->io_spare_cap) { ctx->io_spare[c.cls].push_back(c.h); c.used = true; }
        }
    }
    released_chunks = 0;
    for (const auto &c : X.chunks) if (!c.used) ++released_chunks;
    X.cleanup();                                    // exploration leftovers go back before anything else is allocated
    for (uint32_t role = 0; role < 3 && ok; ++role)
        for (uint32_t i = 0; i < n_bufs && ok; ++i)
            if (bufs[i].role == role && set->maps[i].handles.size() < nch[i]) ok = map_fresh(i);
    if (!ok) { (void)hipGetLastError(); return finish(fail(ctx, IGDSP_ENOMEM, "igdsp_io_alloc: mapping chunks")); }
    for (uint32_t i = 0; i < n_bufs; ++i) bufs[i].ptr = set->maps[i].va;
    if (R.bulk_spread) {
        std::lock_guard<std::mutex> g(ctx->io_mu);
        for (uint32_t i = 0; i < n_bufs; ++i)
            if (bufs[i].role == IGDSP_IO_BULK) ctx->spread_ranges.push_back({(const char *)set->maps[i].va, set->maps[i].bytes});
    }
    if (const char *e = std::getenv("IGDSP_IO_SETTLE")) if (std::atoi(e) == 0) t_ref = 0.f;
    if (t_ref > 0.f) {
        Explorer Y;
        Y.ctx = ctx; Y.s = ctx->stream; Y.probe_n = settle_n;
        const auto w0 = std::chrono::steady_clock::now();
        if (hipEventCreate(&Y.ea) == hipSuccess && hipEventCreate(&Y.eb) == hipSuccess) {
            const float min_wait = std::min(6000.f, (float)((double)released_chunks * (double)chunk / 25e9 * 1e3));
            int quiet = 0;
            for (;;) {
                float t = 0.f;
                if (!Y.time_pair(settle_rd, settle_wr, &t)) break;
                quiet = t <= 1.012f * t_ref ? quiet + 1 : 0;
                const float waited = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
                if (std::getenv("IGDSP_IO_DEBUG")) std::fprintf(stderr, "[igdsp_io] settle: %.4f ms against %.4f before the release (%.0f of >= %.0f ms)\n", t, t_ref, waited, min_wait);
                if ((quiet >= 2 && waited >= min_wait) || waited > 8000.f) break;
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        }
        R.settle_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
        Y.cleanup();
    }
    if (std::getenv("IGDSP_IO_DEBUG") && want_place && R.placed) {   // the same check once more with every exploration chunk released
        Explorer Y;
        Y.ctx = ctx; Y.s = ctx->stream; Y.probe_n = kSrcChunks * (chunk - 4096) / 10240 * 10240;
        if (hipEventCreate(&Y.ea) == hipSuccess && hipEventCreate(&Y.eb) == hipSuccess) {
            int in0 = -1;
            for (uint32_t i = 0; i < n_bufs; ++i) if (bufs[i].role == IGDSP_IO_INPUT && (in0 < 0 || nch[i] > nch[in0])) in0 = (int)i;
            for (uint32_t i = 0; i < n_bufs && in0 >= 0 && nch[in0] >= kSrcChunks; ++i) {
                if (bufs[i].role == IGDSP_IO_INPUT) continue;
                float t = 0.f;
                for (int rep = 0; rep < 3; ++rep)
                    if (Y.time_pair(set->maps[in0].va, set->maps[i].va, &t)) std::fprintf(stderr, "[igdsp_io] after release: input -> buffer %u: %.4f ms\n", i, t);
            }
        }
        Y.cleanup();
    }
    return finish(IGDSP_OK);
}

}  // extern "C"
