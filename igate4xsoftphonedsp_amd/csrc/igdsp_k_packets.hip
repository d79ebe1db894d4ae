// igdsp_k_packets.hip — the packet side: fused depayload + decode + meter (k_meter_rtp64) and ED-137 depayload + gather (k_depayload*).
// Hand-written gfx950 (CDNA4, wave64) kernels; no MFMA: the path is a byte stream with ~4 integer ops per sample, bounded by
// HBM.  Shared device code: igdsp_device.h.
#include "igdsp_device.h"

namespace igdsp {

// ============================================================================
// Fused packet path — k_meter_rtp64: depayload + decode + meter in one pass over 192-byte packet slots
// (include/igdsp.h, igdsp_decode_meter_rtp).  Same machinery as k_meter_chunk64 with 12 pieces per
// slot instead of 10: pieces 0 and 1 of a slot are {size, pad, RTP bytes 0-3} and {RTP bytes 4-19}; they
// run through the pipeline like payload (wasted LUT work on 1/6 of the pieces) but their strip entries
// carry header words instead of partial sums, so the slot's frame lane sees PT / size / ED-137 word at
// fold time and either emits the record or marks the frame EMPTY.  Reads 192 B per frame where the
// two-kernel pipeline (depayload then meter) moves 180 + 160 + 160 + records.
// ============================================================================
constexpr int kSlotPieces = IGDSP_SLOT_BYTES / 16;                 // 12
constexpr int kRtpHalfLoads = kSlotPieces * kChunkFrames / 64;     // 6 loads per lane per 32-slot half
// A frame's 12 strip entries sit in a row of 14 (112 bytes = 28 banks): the fold's six ds_read_b128 per lane then touch 8 distinct
// bank groups per 8-lane pass.  With 12-entry rows (96 bytes = 24 banks) lanes l and l + 4 met in the same banks: the round-2
// counters showed SQ_LDS_BANK_CONFLICT = 4.2e7 cycles per launch (a quarter of the LDS pipe's time) in the packed kernel.
constexpr int kRtpRow = kSlotPieces + 2;
constexpr int kRtpStrip = kSuperFrames * kRtpRow;                  // 896 entries = 7 KiB per wave
constexpr int kRtpWaves = 12;                                      // 64 KiB LUT + 84 KiB strips

// SLOT = true : 192-byte slots (size word + pad + packet at +12), every piece 16-byte aligned.
// SLOT = false: packets packed at `stride` bytes exactly as received; piece addresses are only dword aligned.
// MIXED: payload pieces of the NEXT item's radio packets (bit fr[j] of `nrm`, that item's radio ballot half) sit 8 bytes
// further; the offset is formed at refill time so no per-item offset arrays stay live.
template <bool SLOT, bool MIXED = false>
__device__ __forceinline__ void rtp_half(const uint2 *lut, uint2 *strip_half, uint4 (&d)[kRtpHalfLoads],
                                         const uint32_t am, const uint32_t (&fr)[kRtpHalfLoads], const uint32_t (&pm)[kRtpHalfLoads],
                                         const uint32_t off, const uint32_t lane,
                                         const uint8_t *refill_base, const uint32_t (&roff)[kRtpHalfLoads], const uint32_t nrm = 0u)
{
    uint2 e[2][8];
    uint32_t wa[2], wb[2];
    auto issue = [&](int u) {
        const int j = u >> 1, k = u & 1;
        wa[k] = (u & 1) ? d[j].z : d[j].x;
        wb[k] = (u & 1) ? d[j].w : d[j].y;
        const uint32_t lmj = (uint32_t)__builtin_amdgcn_sbfe(am, fr[j], 1) & 0x80808080u;   // law bit of this piece's packet
        const uint32_t ta = (wa[k] & 0x7F7F7F7Fu) | lmj, tb = (wb[k] & 0x7F7F7F7Fu) | lmj;
        e[k][0] = lut_at(lut, ta, off, 0x0C0C0400u); e[k][1] = lut_at(lut, ta, off, 0x0C0C0500u);
        e[k][2] = lut_at(lut, ta, off, 0x0C0C0600u); e[k][3] = lut_at(lut, ta, off, 0x0C0C0700u);
        e[k][4] = lut_at(lut, tb, off, 0x0C0C0400u); e[k][5] = lut_at(lut, tb, off, 0x0C0C0500u);
        e[k][6] = lut_at(lut, tb, off, 0x0C0C0600u); e[k][7] = lut_at(lut, tb, off, 0x0C0C0700u);
    };
    uint32_t sum = 0, peak = 0, bsum = 0;
    issue(0);
#pragma unroll
    for (int u = 0; u < 2 * kRtpHalfLoads; ++u) {
        const int j = u >> 1, k = u & 1;
        if (u + 1 < 2 * kRtpHalfLoads) issue(u + 1);
        __builtin_amdgcn_sched_barrier(0);
        bsum = __builtin_amdgcn_sad_u8(wa[k], 0u, bsum);
        bsum = __builtin_amdgcn_sad_u8(wb[k], 0u, bsum);
        sum = sum + e[k][0].x + e[k][1].x; sum = sum + e[k][2].x + e[k][3].x;
        sum = sum + e[k][4].x + e[k][5].x; sum = sum + e[k][6].x + e[k][7].x;
        peak = max(max(peak, e[k][0].y), e[k][1].y); peak = max(max(peak, e[k][2].y), e[k][3].y);
        peak = max(max(peak, e[k][4].y), e[k][5].y); peak = max(max(peak, e[k][6].y), e[k][7].y);
        if (k == 1) {
            uint2 ent = make_uint2(sum, peak | (bsum << 16) | probe_fail(d[j], pm[j]));
            // header pieces.  SLOT: piece 0 = slot bytes 0..15 -> {size word, RTP bytes 0-3}; piece 1 = bytes 16..31.
            //                packed: piece 0 = packet bytes 0..15 -> {RTP bytes 0-3, -};      piece 1 = bytes 4..19.
            // either way piece 1 ends with {extension profile/length, ED-137 word}
            const uint32_t hsj = fr[j] >> 8;                    // 0 payload piece, 1 / 2 the two header pieces (fr = slot | hs << 8)
            if (hsj == 1u) ent = SLOT ? make_uint2(d[j].x, d[j].w) : make_uint2(0u, d[j].x);
            if (hsj == 2u) ent = make_uint2(d[j].z, d[j].w);
            strip_half[j * 64 + lane + (uint32_t)(kRtpRow - kSlotPieces) * (fr[j] & 31u)] = ent;   // piece p of the half -> row p / 12, column p % 12
            // SLOT: piece p of the half sits at p * 16 (roff[0] = this lane's piece 0; no per-piece offset registers)
            const uint32_t ro = SLOT ? roff[0] + (uint32_t)j * 1024u : roff[j] + ((MIXED && hsj == 0u) ? ((nrm >> (fr[j] & 31u)) & 1u) * 8u : 0u);
            d[j] = SLOT ? ld_stream(reinterpret_cast<const uint4 *>(refill_base + ro)) : ld16_dw(refill_base + ro);
            sum = 0; peak = 0; bsum = 0;
        }
    }
}

// Short payloads in the fused packet kernels.  transport_rtp_cb accepts any payloadlen = size - header
// (TransportAdapter.cpp:270-291; the hook anticipates 164 and 24, roip_ed137.cpp:6561-6562).  The pipeline above is built
// for whole 160-byte payloads; a packet with 0 < payloadlen < 160 is rare, so its FRAME LANE re-meters it alone, straight
// from the packet (L2-hot or re-fetched), with the same LUT: one dword per step, bytes past `len` masked out of every
// accumulator.  Reads stay inside the packet's slot (len < 160 <= slot payload).  Costs ~25 instructions per dword in the
// lanes that need it and one wave-uniform branch for everybody else.
__device__ __forceinline__ void meter_short(const uint2 *lut, const uint32_t off, const uint8_t *pp, const uint32_t len, const bool alaw,
                                            uint64_t &s, uint32_t &peak, uint32_t &bsum, bool &probe)
{
    const uint32_t lm = alaw ? 0x80808080u : 0u;
    uint64_t acc = 0;
    uint32_t pk = 0, bs = 0, fail = 0;
    for (uint32_t i = 0; i < len; i += 4u) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(pp + i);
        const uint32_t nv = min(len - i, 4u);
        const uint32_t t = (w & 0x7F7F7F7Fu) | lm;
        const uint2 e0 = lut_at(lut, t, off, 0x0C0C0400u), e1 = lut_at(lut, t, off, 0x0C0C0500u);
        const uint2 e2 = lut_at(lut, t, off, 0x0C0C0600u), e3 = lut_at(lut, t, off, 0x0C0C0700u);
        acc += (uint64_t)(e0.x + (nv > 1u ? e1.x : 0u) + (nv > 2u ? e2.x : 0u) + (nv > 3u ? e3.x : 0u));   // 4 x 2^26 fits u32
        pk = max(max(pk, e0.y), max(nv > 1u ? e1.y : 0u, max(nv > 2u ? e2.y : 0u, nv > 3u ? e3.y : 0u)));
        bs = __builtin_amdgcn_sad_u8(nv >= 4u ? w : (w & ((1u << (8u * nv)) - 1u)), 0u, bs);
        // the reference's silence probe: payload bytes 28 / 38 / 48
        if (i == 28u || i == 48u) fail |= (w ^ 0xD5u) & 0xFFu;
        if (i == 36u) fail |= ((w >> 16) ^ 0xD5u) & 0xFFu;
    }
    s = acc; peak = pk; bsum = bs; probe = len > 48u && fail == 0u;
}

// WIN == 2 helpers: one frame's effect on a channel's consecutive-silence run (adapter->rtpFalse, TransportAdapter.cpp:657-673), and
// the commit of every consecutive frame whose masks are in group gj's ring, by the wave that finds the group free.
__device__ __forceinline__ void win_step(uint32_t &run, uint32_t &al, bool pr, bool npr, uint32_t alarm)
{
    run = npr ? 0u : run + (pr ? 1u : 0u);
    al += (pr && run == alarm) ? 1u : 0u;
}

__device__ __forceinline__ void win_drain(uint32_t *commit, const uint32_t *flag, const uint4 *ring, uint32_t *runs, uint32_t *alarms, uint32_t gj, uint32_t lane, uint32_t alarm)
{
    constexpr uint32_t kLock = 0x80000000u;
    auto ldu = [&](const uint32_t *p) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); };
    auto slot = [&](uint32_t f) { return gj * (uint32_t)kWinRing + (f & (uint32_t)(kWinRing - 1)); };
    uint32_t *const S = commit + gj;
    const uint32_t cl = gj * 64u + lane;
    uint32_t n = ldu(S);
    bool more = (n & kLock) == 0u;                            // held: the next frame that comes this way looks again
    while (more && ldu(&flag[slot(n)]) == n + 1u) {           // the next frame to commit is there: take the group if it is still free
        uint32_t o = n;
        if (lane == 0u) __hip_atomic_compare_exchange_strong(S, &o, n | kLock, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t run = runs[cl], al = alarms[cl];
        o = (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
        if (o != n) { more = (o & kLock) == 0u; n = o; continue; }     // somebody else moved the group on (or holds it)
        do {
            const uint2 m = reinterpret_cast<const uint2 *>(&ring[slot(n)])[lane >> 5];
            win_step(run, al, ((m.x >> (lane & 31u)) & 1u) != 0u, ((m.y >> (lane & 31u)) & 1u) != 0u, alarm);
            n += 1u;
        } while (ldu(&flag[slot(n)]) == n + 1u);
        runs[cl] = run; alarms[cl] = al;
        if (lane == 0u) __hip_atomic_store(S, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// WIN == 2, one folded frame of one channel (this lane): the window's LDS atomics and the run's in-order commit.
__device__ __forceinline__ void win_blk_item(const WinArgs &win, uint32_t id_cur, uint32_t lane, uint32_t l, bool metered, uint32_t ed, uint64_t sumsq16,
                                             uint32_t peak, uint32_t bm, uint32_t fl, uint32_t c_seen, uint32_t c_run, uint32_t c_al,
                                             uint32_t *wst, uint32_t *w_commit, uint32_t *w_flag, uint4 *w_ring, uint32_t *w_run, uint32_t *w_alarms)
{
    const uint32_t gj = id_cur & (win.gpb - 1u), cl = gj * 64u + lane;          // this lane's channel within the block
    const bool fold = metered && (win.gate_mask == 0u || (ed & win.gate_mask) != 0u);
    if (fold) {                               // integer sums / max / min: any order
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(wst) + cl, (unsigned long long)sumsq16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(wst + 2 * kWinBlkCh + cl, 1u | ((fl & IGDSP_FLAG_SILENT) ? 0x100u : 0u) | ((fl & IGDSP_FLAG_CLIPPED) ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(wst + 3 * kWinBlkCh + cl, bm | (l << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(wst + 4 * kWinBlkCh + cl, peak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(wst + 5 * kWinBlkCh + cl, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_min(wst + 6 * kWinBlkCh + cl, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifdef IGDSP_BLK_NORUN            // A/B builds only (wrong runs on purpose): what the commit protocol costs
    if (win.probe != nullptr && l == 77777u) {
#else
    if (win.probe != nullptr) {
#endif
        // The run moves in FRAME order and the frames of a group are folded by different waves in any order.  Nobody
        // waits (a first version made every frame wait for its predecessor: +3 %), and the common case has no
        // LDS round trip in the wave's way (with two the launch was 3 % slower: an item's time is its loads' latency).
        // w_commit[j] = the next frame of group j to commit, bit 31 = a wave is committing.  The compare-and-swap
        // issued at the top of the fold takes the group if this frame IS the next one: the frame is applied from
        // registers and the group released, nothing read back.  Any other wave leaves its frame's probe / reset masks
        // in the group's ring and then looks at the group: if it is free and its next frame is there it takes the group
        // and applies every consecutive frame it finds.  Frames left while a wave held the group stay until a later
        // frame finds the group free — that frame cannot be the group's next one, so it comes this way; what is left
        // when the block's items are through is applied at the block's end.
        const bool valid = metered && l > 48u, pr = valid && (fl & IGDSP_FLAG_PROBE_D5) != 0u, npr = valid && !pr;
        const uint32_t fr_no = id_cur >> win.gsh, kLock = 0x80000000u;
        uint32_t run = c_run, al = c_al;
        uint32_t seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_seen);
        if (seen == fr_no) {                 // this frame was next and the group free
            win_step(run, al, pr, npr, win.alarm);
            w_run[cl] = run; w_alarms[cl] = al;
            if (lane == 0u) __hip_atomic_store(&w_commit[gj], fr_no + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            while ((seen & ~kLock) + (uint32_t)kWinRing <= fr_no) {     // (the slot's previous frame, 16 back: never pending in practice)
                __builtin_amdgcn_s_sleep(1);
                win_drain(w_commit, w_flag, w_ring, w_run, w_alarms, gj, lane, win.alarm);
                seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&w_commit[gj], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
            }
            const uint64_t mp = __ballot(pr), mn = __ballot(npr);
            if (lane == 0u) {
                const uint32_t sl = gj * (uint32_t)kWinRing + (fr_no & (uint32_t)(kWinRing - 1));
                w_ring[sl] = make_uint4((uint32_t)mp, (uint32_t)mn, (uint32_t)(mp >> 32), (uint32_t)(mn >> 32));
                __hip_atomic_store(&w_flag[sl], fr_no + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            win_drain(w_commit, w_flag, w_ring, w_run, w_alarms, gj, lane, win.alarm);     // (skipped when the group was seen held: +0.7 %, the rings fill)
        }
    }
}

// MIXED (packed form only): the header length is per channel, 20 bytes where radio[c] != 0 and 12 elsewhere (SIP and
// ED-137 legs in one launch, as in the reference's process); `hdr` is then ignored.  The radio flags travel like the
// codec ids: the frame lanes fetch them one item ahead and a ballot hands every piece its packet's bit.
//
// WIN (igdsp_decode_meter_window; C % 64 == 0): the ED-137 gated window in the same pass.  With C a multiple of 64, super-chunk
// sidx = f * (C / 64) + g is frame f of channel group g, and lane l is channel 64 g + l in every one of them — so the kernel
// only changes the ORDER in which a wave visits super-chunks: a work unit = (group g, segment of the F frames), walked in frame
// order like k_roundtrip_lut64 walks its items, with the window (keeplogAudioLevel's count / sum / max / min, Functions.cpp:
// 2126-2145, plus sum of squares, peak-hold, silent / clipped counts) and the consecutive-silence run (adapter->rtpFalse,
// TransportAdapter.cpp:657-673) of each channel in the lane's registers.  The per-frame gate comes from the ED-137 word the frame
// lane parses anyway (masks Functions.cpp:1136, 1160).  At a unit's last frame the window merges into hold[c] by integer
// atomics would be possible (exact, order-free) but all units end together and the burst costs 12 % of the launch; instead the
// unit's window {sum of squares, frames, byte-mean sum | samples, peak-hold | level-max, level-min, silent | clipped} and its run
// {probe frames before the first reset, run at the end, alarms after the first reset, had a reset} leave as three 16-byte words in
// work[seg][k][c], which k_window_finish folds into hold[c] / probe[c] in segment order (integer sums, max, min: bit-identical
// to the sequential fold).  The id stream of a wave carries the unit's segment
// (bits 27-29) and a last-frame-of-unit flag (bit 31), so the load pipeline runs straight across unit boundaries.
//
// WIN == 2 (the default where the shape allows, igdsp_capi.hip): the state moves from the wave to the BLOCK.  Block b owns the
// gpb = 4, 2 or 1 consecutive channel groups b * gpb ... for the whole launch and hands their F * gpb items to its waves one at a
// time in (frame, group) order from an LDS counter — the item-level balance of the time-major kernels inside a block, and the
// blocks walk the frames roughly together, so the memory system sees the time-major order too.  The windows of the block's
// <= 256 channels live in LDS and move by LDS atomics (integer add / max / min: any order; measured free); the run is
// order-dependent, so a group's frames commit in frame order through a 16-slot ring of probe / reset masks and a lock, the
// holder applying every consecutive frame that is there — nobody waits.  At its end the block folds its windows into hold[c]
// and writes probe[c] itself: no summaries, no finish kernel.  hold[c] is fetched one item before the end (a global load at the
// block's end waits ~5 us behind the other blocks' streams).  Packed LDS counters: F <= 255 per launch (the launcher splits).
template <bool AGG, bool SLOT, bool MIXED = false, int WIN = 0>
__global__ __launch_bounds__(kRtpWaves * 64) void k_meter_rtp64(
    const uint8_t *__restrict__ slots, const uint16_t *__restrict__ sizes, const uint8_t *__restrict__ codec, uint32_t C,
    uint32_t n_frames, uint32_t stride, uint32_t hdr, igdsp_frame_stats *__restrict__ stats, igdsp_rtp_info *__restrict__ info,
    igdsp_aggregate *agg, uint32_t rank, uint32_t *gqueue, const uint8_t *__restrict__ radio = nullptr, const WinArgs win = WinArgs{})
{
    static_assert(!(SLOT && MIXED), "slots always hold 20-byte headers");
    if (MIXED) hdr = 12u;
    __shared__ __attribute__((aligned(16))) uint2 lds[kLutEntries + kRtpWaves * kRtpStrip];
    __shared__ BlockQueue<kRtpWaves> bq;
    __shared__ AggBlock aggb;
    // WIN == 2: the windows of the block's own channels {sum of squares (2 dwords), frames | silent << 8 | clipped << 16,
    // byte-mean sum | samples << 16, peak-hold, level max, level min} x kWinBlkCh channels, moved by LDS atomics
    __shared__ uint32_t wst[WIN == 2 ? 7 * kWinBlkCh : 1];
    __shared__ uint32_t w_ticket;                        // WIN == 2: order in which the waves start their last item
    // WIN == 2: the consecutive-silence run is order-dependent and the frames of a channel are folded by different waves: a group's
    // frames COMMIT in frame order (w_commit[j] = next frame of group j to commit), the run and the alarms of the block's channels
    // live here between the launch's start and end
    __shared__ uint32_t w_run[WIN == 2 ? kWinBlkCh : 1], w_alarms[WIN == 2 ? kWinBlkCh : 1], w_commit[4];
    __shared__ __attribute__((aligned(16))) uint4 w_ring[WIN == 2 ? 4 * kWinRing : 1];       // probe / reset masks of frames not yet committed
    __shared__ uint32_t w_flag[WIN == 2 ? 4 * kWinRing : 1];                                  // frame + 1 once the slot holds that frame's masks
    uint32_t gb1 = 0;
#ifdef IGDSP_BLK_STAMP
    const uint64_t t_stamp0 = wall_clock64();
#endif
    if (threadIdx.x == 0 && gqueue != nullptr) gb1 = atomicAdd(gqueue, 1u);
    igdsp_chan_probe p0 = igdsp_chan_probe{0u, 0u};              // WIN == 2: the channels' runs so far (the load's latency hides under the table fill)
    if (WIN == 2 && threadIdx.x < win.gpb * 64u && win.probe != nullptr) p0 = win.probe[blockIdx.x * win.gpb * 64u + threadIdx.x];
    fill_lut(lds);
    if (threadIdx.x == 0) { bq_init(bq, gqueue, gridDim.x, gb1); agg_block_init(aggb); if (WIN == 2) { bq.next = 0u; w_ticket = 0u; } }
    if (WIN == 2) {
        for (uint32_t i = threadIdx.x; i < 7u * (uint32_t)kWinBlkCh; i += blockDim.x) wst[i] = i >= 6u * (uint32_t)kWinBlkCh ? 255u : 0u;
        if (threadIdx.x < win.gpb * 64u) { w_run[threadIdx.x] = p0.run; w_alarms[threadIdx.x] = p0.alarms; }
        if (threadIdx.x < 4u) w_commit[threadIdx.x] = 0u;
        if (threadIdx.x < 4u * (uint32_t)kWinRing) w_flag[threadIdx.x] = 0u;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __syncthreads();

    uint2 *strip = lds + kLutEntries + wave * kRtpStrip;
    const uint32_t off = (lane & 31u) * 8u;
    uint32_t pm[kRtpHalfLoads], fr[kRtpHalfLoads];        // fr = slot within the half | header-piece selector << 8 (shifts and v_bfe only look at the low 5 bits)
    uint32_t roff0[kRtpHalfLoads], roff1[kRtpHalfLoads];       // byte offset of this lane's pieces inside a super-chunk
#pragma unroll
    for (int j = 0; j < kRtpHalfLoads; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane;
        const uint32_t f = p / 12u;                       // slot within the 32-slot half
        const uint32_t q = p - f * 12u;                   // piece within the slot: 0, 1 header; 2..11 payload
        fr[j] = f | ((q < 2u ? q + 1u : 0u) << 8);
        pm[j] = q < 2u ? 0u : probe_mask(q - 2u);
        if (SLOT) {
            roff0[j] = p * 16u;
            roff1[j] = (p + (uint32_t)(kChunkFrames * kSlotPieces)) * 16u;      // the second half's pieces follow the first half's 384
        } else {
            const uint32_t po = q == 0u ? 0u : (q == 1u ? 4u : hdr + 16u * (q - 2u));
            roff0[j] = f * stride + po;
            roff1[j] = (f + (uint32_t)kChunkFrames) * stride + po;
        }
    }
    const uint32_t G = gridDim.x;
    const uint32_t n_super = n_frames / kSuperFrames;
    const uint64_t super_bytes = (uint64_t)kSuperFrames * (SLOT ? (uint32_t)IGDSP_SLOT_BYTES : stride);
    // launch-aggregate partials: per lane sum of squares, samples (< 2^32 per lane and launch), byte-mean sum, peak; the three
    // counts are wave-uniform popcounts of ballots and live in SGPRs
    uint64_t a_sumsq = 0;
    uint32_t a_samp = 0, a_bm = 0, a_peak = 0;
    uint32_t u_frames = 0, u_sil = 0, u_clip = 0;

    auto fetch_radio = [&](uint32_t sidx) { return MIXED ? (uint32_t)radio[(sidx * (uint32_t)kSuperFrames + lane) % C] : 0u; };
    // byte offsets of this lane's pieces for an item whose packets' radio bits are rm: payload pieces of radio packets sit 8 bytes further
    auto offsets = [&](uint64_t rm, uint32_t (&o0)[kRtpHalfLoads], uint32_t (&o1)[kRtpHalfLoads]) {
#pragma unroll
        for (int j = 0; j < kRtpHalfLoads; ++j) {
            const uint32_t f = fr[j] & 31u, b0 = (uint32_t)(rm >> f) & 1u, b1 = (uint32_t)(rm >> (f + 32u)) & 1u;
            o0[j] = roff0[j] + (fr[j] < 256u ? 8u * b0 : 0u);
            o1[j] = roff1[j] + (fr[j] < 256u ? 8u * b1 : 0u);
        }
    };
    auto ld = [&](const uint8_t *b, uint32_t o) { return SLOT ? ld_stream(reinterpret_cast<const uint4 *>(b + o)) : ld16_dw(b + o); };
    auto fetch_pt = [&](uint32_t sidx) { return (uint32_t)codec[(sidx * (uint32_t)kSuperFrames + lane) % C]; };
    const uint32_t n_batches = IGDSP_SPREAD_METER ? (n_super + (uint32_t)kRtpWaves - 1u) / (uint32_t)kRtpWaves : 0u;   // records only: no spreading
    // WIN: the wave's id stream = its units (round r: unit r * total_waves + wave * G + block, the roundtrip kernels' "a grid
    // apart" order), each walked in frame order; wave-uniform state in SGPRs
    // (kept minimal: the packed-packet kernel already runs at the rate of a compute-free kernel with its access pattern, so
    // every instruction added per item shows — a first version with 64-bit unit arithmetic and the unit's frame bounds live across
    // the loop executed twice the scalar and 10 % more vector instructions and ran 10 % slower)
    uint32_t w_next = 0, w_left = 0, w_seg = 0, w_unit = wave * G + blockIdx.x;          // next id, frames left in the unit, its segment, next unit
    const uint32_t nw = WIN ? (uint32_t)(blockDim.x >> 6) : (uint32_t)kRtpWaves;       // WIN blocks may run fewer waves than the strips allow
    const uint32_t w_units = win.n_groups * win.n_seg, w_stride = G * nw;                // (< 2^27: the launcher)
    const uint32_t b_items = win.F * win.gpb, b_first = blockIdx.x * win.gpb;                 // WIN == 2: the block's own items, in (frame, group) order
    auto grab = [&]() -> uint32_t {
        if (!WIN) return bq_grab(bq, gqueue, G, lane, n_batches);
        if (WIN == 2) {
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&bq.next, 1u);
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            return v < b_items ? v : 0xFFFFFFFFu;            // item v of the block = frame v >> gsh of its group v & (gpb - 1)
        }
#ifdef IGDSP_WIN_ASC              // A/B builds only (wrong windows on purpose): the WIN code over the ascending static item order
        { const uint32_t id = w_unit; w_unit += w_stride; return id < n_super ? id : 0xFFFFFFFFu; }
#endif
        if (w_left == 0u) {
            if (w_unit >= w_units) return 0xFFFFFFFFu;
            w_seg = w_unit / win.n_groups;
            const uint32_t f_lo = (uint32_t)(((uint64_t)win.F * w_seg) / win.n_seg);
            w_left = (uint32_t)(((uint64_t)win.F * (w_seg + 1u)) / win.n_seg) - f_lo;     // >= 1: n_seg <= F (the launcher)
            w_next = f_lo * win.n_groups + (w_unit - w_seg * win.n_groups);
            w_unit += w_stride;
        }
        const uint32_t id = w_next | (w_seg << 27) | (w_left == 1u ? 0x80000000u : 0u);
        w_next += win.n_groups;
        w_left -= 1u;
        return id;
    };
    auto id_ok = [&](uint32_t id) { return WIN ? id != 0xFFFFFFFFu : id < n_super; };
    auto id_sidx = [&](uint32_t id) { return WIN == 2 ? (id >> win.gsh) * win.n_groups + b_first + (id & (win.gpb - 1u)) : (WIN ? (id & 0x07FFFFFFu) : id); };
    // the window of the unit under way (this lane = one channel): sum of squares, frames, byte-mean sum, samples,
    // {peak_hold | level_max << 16}, level_min, {n_silent | n_clipped << 16}; the silence run {run so far, probe frames before the
    // first reset | had-a-reset << 31, alarms after the first reset}
    uint64_t w_sumsq = 0;
    uint32_t w_cnt = 0, w_lsum = 0, w_samp = 0, w_pm = 0, w_min = 255u, w_sc = 0;
    uint32_t r_trail = 0, r_lead = 0, r_hits = 0;

    // WIN == 2: the fold at the block's end needs hold[c] / probe[c] / gate[c] of the block's channels, and a global load issued
    // there waits ~5 us behind the other blocks' streams (measured).  So the first gpb waves to START their last item (or to find
    // they have none) fetch one group's words each, a whole item ahead of the fold, and merge that group later.
    igdsp_chan_hold e_hold = igdsp_chan_hold{};
    uint32_t e_ticket = 0xFFFFFFFFu;
    bool e_open = false;
    auto end_prefetch = [&]() {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&w_ticket, 1u);
        e_ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        if (e_ticket < win.gpb) {
            const uint32_t c = (b_first + e_ticket) * 64u + lane;
            e_hold = win.hold[c];
            e_open = win.gate == nullptr || win.gate[c] != 0;
        }
    };
    uint32_t id_cur = WIN ? grab() : spread_batch(blockIdx.x, n_batches) * (uint32_t)kRtpWaves + wave;     // batch blockIdx.x, slot = wave
    uint32_t sidx = id_sidx(id_cur);
    if (id_ok(id_cur)) {
        uint4 X[kRtpHalfLoads], Y[kRtpHalfLoads];
        uint32_t cur_pt = fetch_pt(sidx);
        uint32_t cur_radio = fetch_radio(sidx);
        {
            const uint8_t *b0 = slots + (uint64_t)sidx * super_bytes;
            if (MIXED) {
                uint32_t o0[kRtpHalfLoads], o1[kRtpHalfLoads];
                offsets(__ballot(cur_radio != 0u), o0, o1);
#pragma unroll
                for (int j = 0; j < kRtpHalfLoads; ++j) X[j] = ld(b0, o0[j]);
#pragma unroll
                for (int j = 0; j < kRtpHalfLoads; ++j) Y[j] = ld(b0, o1[j]);
            } else {
#pragma unroll
                for (int j = 0; j < kRtpHalfLoads; ++j) X[j] = ld(b0, roff0[j]);
#pragma unroll
                for (int j = 0; j < kRtpHalfLoads; ++j) Y[j] = ld(b0, roff1[j]);
            }
        }
        uint32_t s_next = grab();
        for (;;) {
            const bool has_next = id_ok(s_next);
            if (WIN == 2 && !has_next) end_prefetch();            // (once: the loop ends with this item)
            const uint32_t s_load = has_next ? id_sidx(s_next) : 0u;
            const uint32_t f0 = sidx * kSuperFrames;
            const bool my_alaw = cur_pt == IGDSP_PT_PCMA;
            const uint64_t amask = __ballot(my_alaw);
            const uint32_t am_lo = (uint32_t)amask, am_hi = (uint32_t)(amask >> 32);
            const uint8_t *nbase = slots + (uint64_t)s_load * super_bytes;
            const uint32_t nxt_pt = fetch_pt(s_load);
            const uint32_t nxt_radio = fetch_radio(s_load);
            const uint32_t hbytes = SLOT ? 20u : (MIXED ? (cur_radio != 0u ? 20u : 12u) : hdr);   // of this lane's own packet
            const uint32_t full = hbytes + (uint32_t)kFrame;
            uint32_t my_size = full;
            if (!SLOT && sizes != nullptr) my_size = sizes[f0 + lane];
            if (MIXED) {
                const uint64_t nrm = __ballot(nxt_radio != 0u);
                rtp_half<SLOT, true>(lds, strip, X, am_lo, fr, pm, off, lane, nbase, roff0, (uint32_t)nrm);
                rtp_half<SLOT, true>(lds, strip + kRtpStrip / 2, Y, am_hi, fr, pm, off, lane, nbase, roff1, (uint32_t)(nrm >> 32));
            } else {
                rtp_half<SLOT>(lds, strip, X, am_lo, fr, pm, off, lane, nbase, roff0);
                rtp_half<SLOT>(lds, strip + kRtpStrip / 2, Y, am_hi, fr, pm, off, lane, nbase, roff1);
            }
            const uint32_t s_after = has_next ? grab() : 0xFFFFFFFFu;   // its LDS round trip hides under the fold below
            wave_lds_fence();
            {
                // WIN == 2: the try for the group's commit word and the run state go out here and come back under the fold
                // (issued after the row reads instead, with less time holding the group: 0.2773-0.2793 ms against 0.2726-0.2765; a plain
                // load here and a plain store at the end — only the frame the word names can commit next, so it needs no atomic —
                // measured 2.6 % over the kernel without run tracking where this form measures 1.4 %: a frame that arrives while its
                // predecessor is being folded sees the group held and leaves after one look)
                uint32_t c_seen = 0, c_run = 0, c_al = 0;
                if (WIN == 2 && win.probe != nullptr) {
                    const uint32_t fr_no = id_cur >> win.gsh, gj = id_cur & (win.gpb - 1u);
                    c_seen = fr_no;
                    if (lane == 0u) __hip_atomic_compare_exchange_strong(&w_commit[gj], &c_seen, fr_no | 0x80000000u, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    c_run = w_run[gj * 64u + lane]; c_al = w_alarms[gj * 64u + lane];    // (after the compare-and-swap in LDS order: if it took the group these are the last holder's stores)
                }
                const uint4 *row = reinterpret_cast<const uint4 *>(strip + lane * kRtpRow);       // 112-byte rows, 16-byte aligned
                const uint4 h = row[0];                   // {size word | 0, RTP bytes 0-3, ext profile/length, ED-137 word}
                if (!SLOT) asm volatile("" ::"v"(h.x));     // packed: h.x is unused, and without this the row is fetched as eleven dword PAIRS starting at
                                                          // dword 1 (ds_read2_b32: 4-way bank conflicts at the 112-byte row stride) instead of six ds_read_b128
                uint64_t s = 0;
                uint32_t peak = 0, bsum = 0, fail = 0;
#pragma unroll
                for (int i = 1; i < kSlotPieces / 2; ++i) {
                    const uint4 v = row[i];
                    s += (uint64_t)(v.x + v.z);
                    peak = max(max(peak, v.y & 0x7FFFu), v.w & 0x7FFFu);
                    bsum += ((v.y >> 16) & 0x7FFFu) + ((v.w >> 16) & 0x7FFFu);
                    fail |= v.y | v.w;
                }
                // header: same rules as parse_rtp()
                const uint32_t size = SLOT ? (h.x & 0xFFFFu) : my_size, w0 = h.y, pt = (w0 >> 8) & 0x7Fu;
                uint32_t hf = (((w0 >> 6) & 3u) == 2u ? IGDSP_RTP_V2 : 0u) | ((w0 & 0x10u) ? IGDSP_RTP_X : 0u) |
                              ((w0 & 0x8000u) ? IGDSP_RTP_MARKER : 0u);
                uint32_t ed = 0, plen = 0;
                if (size < hbytes) hf = IGDSP_RTP_RUNT;
                else {
                    plen = size - hbytes;
                    if (hbytes == 20u) {
                        if (pt == 8u || pt == 0u || pt == 18u || pt == 123u) ed = __builtin_bswap32(h.w);
                        if ((w0 & 0x10u) && h.z == 0x01006701u) hf |= IGDSP_RTP_ED137_OK;
                    }
                    if (pt == 123u) hf |= IGDSP_RTP_KEEPALIVE;
                    if (plen > (uint32_t)kFrame) hf |= IGDSP_RTP_OVERSIZE;
                    else if ((pt == 0u || pt == 8u) && plen > 0u) hf |= IGDSP_RTP_METERED;
                }
                const bool pt_ok = pt == cur_pt && (pt == 0u || pt == 8u);
                const bool whole = size == full && pt_ok;
                const bool shortp = pt_ok && size > hbytes && size < full;           // 0 < payloadlen < 160
                const bool metered = whole || shortp;
                const uint32_t fi = f0 + lane;
                if (info != nullptr) {
                    uint2 rec;
                    rec.x = ed;
                    rec.y = (size < hbytes ? 0u : plen) | ((size >= 2u ? pt : 0u) << 16) | (hf << 24);
                    *reinterpret_cast<uint2 *>(info + fi) = rec;
                }
                uint32_t bm = 0, fl = 0;
                uint4 rec = make_uint4(0u, 0u, 0u, (uint32_t)IGDSP_FLAG_EMPTY << 24);
                if (whole) rec = pack_stats160(s, peak, bsum, my_alaw, (fail >> 31) == 0u, bm, fl);
                if (__ballot(shortp) != 0ull) {           // wave-uniform, rare: see meter_short
                    if (shortp) {
                        const uint8_t *pp = slots + (uint64_t)sidx * super_bytes + lane * (SLOT ? (uint32_t)IGDSP_SLOT_BYTES : stride) +
                                            (SLOT ? (uint32_t)IGDSP_SLOT_PAYLOAD_OFFSET : hbytes);
                        bool pr;
                        meter_short(lds, off, pp, plen, my_alaw, s, peak, bsum, pr);
                        rec = pack_stats(s << 4, peak, bsum, plen, my_alaw, pr, bm, fl);
                    }
                }
                if (!WIN || stats != nullptr) st_stream(reinterpret_cast<uint4 *>(stats + fi), rec);   // (WIN: the records are optional — a host that only wants the windows)
                if (AGG) {
                    if (metered) { a_sumsq += s << 4; a_samp += whole ? (uint32_t)kFrame : plen; a_bm += bm; a_peak = max(a_peak, peak); }
                    u_frames += (uint32_t)__builtin_popcountll(__ballot(metered));
                    u_sil += (uint32_t)__builtin_popcountll(__ballot(metered && (fl & IGDSP_FLAG_SILENT) != 0u));
                    u_clip += (uint32_t)__builtin_popcountll(__ballot(metered && (fl & IGDSP_FLAG_CLIPPED) != 0u));
                }
                if (WIN == 2) {
                    win_blk_item(win, id_cur, lane, whole ? (uint32_t)kFrame : plen, metered, ed, s << 4, peak, bm, fl, c_seen, c_run, c_al,
                                 wst, w_commit, w_flag, w_ring, w_run, w_alarms);
                } else if (WIN) {
#ifndef IGDSP_WIN_NOBOOK          // A/B builds only: the walk without the per-frame bookkeeping (wrong windows on purpose)
                    // branch-free: every step is a select on the lane's own predicates
                    const uint32_t l = whole ? (uint32_t)kFrame : plen;
                    // consecutive-silence run: only a metered frame that holds the probe bytes (payload length > 48) moves it
                    const bool valid = metered && l > 48u, pr = valid && (fl & IGDSP_FLAG_PROBE_D5) != 0u, npr = valid && !pr;
                    const uint32_t nt = r_trail + (pr ? 1u : 0u);
                    r_hits += (pr && (r_lead >> 31) != 0u && nt == win.alarm) ? 1u : 0u;
                    r_lead = (npr && (r_lead >> 31) == 0u) ? (r_trail | 0x80000000u) : r_lead;
                    r_trail = npr ? 0u : nt;
                    // frame gate from the ED-137 word of this frame's own packet: win.gate_mask = SQU bit 28 and / or PTT type bits
                    // 31-29 (Functions.cpp:1160, 1136); 0 = every metered frame
                    const bool fold = metered && (win.gate_mask == 0u || (ed & win.gate_mask) != 0u);
                    w_sumsq += fold ? (s << 4) : 0ull;
                    w_cnt += fold ? 1u : 0u; w_lsum += fold ? bm : 0u; w_samp += fold ? l : 0u;
                    w_pm = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2u16_t, w_pm), __builtin_bit_cast(v2u16_t, fold ? (peak | (bm << 16)) : 0u)));
                    w_min = min(w_min, fold ? bm : 255u);
                    w_sc += fold ? (((fl & IGDSP_FLAG_SILENT) ? 1u : 0u) + ((fl & IGDSP_FLAG_CLIPPED) ? 0x10000u : 0u)) : 0u;
#endif
                    if ((id_cur >> 31) != 0u) {             // wave-uniform: last frame of the unit
                        // the unit's window and run leave as three 16-byte words per channel, work[seg][k][c] (64 lanes = 1 KiB per
                        // store); k_window_finish chains the segments of a channel in order.  (A first version merged the
                        // windows into hold[c] with device atomics at this point: all units end together, and the burst of
                        // 7 atomics x 64 lanes x 3 072 waves cost the launch 12 %.)
                        const uint32_t cme = (sidx % win.n_groups) * (uint32_t)kSuperFrames + lane, seg = (id_cur >> 27) & 7u;
                        uint4 *wk = win.work + ((uint64_t)seg * 3u * C + cme);
                        wk[0] = make_uint4((uint32_t)w_sumsq, (uint32_t)(w_sumsq >> 32), w_cnt, w_lsum);
                        wk[C] = make_uint4(w_samp, w_pm, w_min, w_sc);
                        wk[2u * (uint64_t)C] = make_uint4((r_lead >> 31) ? (r_lead & 0x7FFFFFFFu) : r_trail, r_trail, r_hits, r_lead >> 31);
                        w_sumsq = 0; w_cnt = 0; w_lsum = 0; w_samp = 0; w_pm = 0; w_min = 255u; w_sc = 0;
                        r_trail = 0; r_lead = 0; r_hits = 0;
                    }
                }
            }
            wave_lds_fence();
            if (!has_next) break;
            id_cur = s_next;
            sidx = id_sidx(s_next);
            s_next = s_after;
            cur_pt = nxt_pt;
            cur_radio = nxt_radio;
        }
    }
    if (WIN == 2) {
        // Every item of the block's channels has been handed out; when all waves are here they have been folded and committed.
        // The gpb waves holding hold[c] of a group merge its window (the block owns the channel for the launch: plain
        // read-modify-write) and write the run back.
        if (e_ticket == 0xFFFFFFFFu) end_prefetch();           // a wave that never had an item
        const bool mine = e_ticket < win.gpb;
        const uint32_t tch = (mine ? e_ticket : 0u) * 64u + lane, c = b_first * 64u + tch;
        __syncthreads();
#ifdef IGDSP_BLK_STAMP
        const uint64_t t_loop = wall_clock64();
#endif
        if (win.probe != nullptr) {                          // frames left in the rings while their group was held: nobody holds a group now
            if (wave < win.gpb) win_drain(w_commit, w_flag, w_ring, w_run, w_alarms, wave, lane, win.alarm);
            __syncthreads();
        }
        if (mine) {
            if (e_open) {
                const uint32_t wa = wst[2 * kWinBlkCh + tch], wb = wst[3 * kWinBlkCh + tch];
                if ((wa & 0xFFu) != 0u) {
                    igdsp_chan_hold h = e_hold;
                    h.sumsq_acc += reinterpret_cast<const unsigned long long *>(wst)[tch]; h.count += wa & 0xFFu; h.level_sum += wb & 0xFFFFu; h.samples += wb >> 16;
                    h.peak_hold = (uint16_t)max((uint32_t)h.peak_hold, wst[4 * kWinBlkCh + tch]);
                    h.level_max = (uint8_t)max((uint32_t)h.level_max, wst[5 * kWinBlkCh + tch]);
                    h.level_min = (uint8_t)min((uint32_t)h.level_min, wst[6 * kWinBlkCh + tch]);
                    h.n_silent += (wa >> 8) & 0xFFu; h.n_clipped += (wa >> 16) & 0xFFu;
                    win.hold[c] = h;
                }
            }
            if (win.probe != nullptr) win.probe[c] = igdsp_chan_probe{w_run[tch], w_alarms[tch]};
        }
#ifdef IGDSP_BLK_STAMP
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            win.work[(uint64_t)gridDim.x * b_items + blockIdx.x] = make_uint4((uint32_t)t_stamp0, (uint32_t)t_loop, (uint32_t)wall_clock64(), xcc);
        }
#endif
    }
    wave_exit(AGG ? agg : nullptr, rank, aggb, nw, lane, gqueue, G, a_sumsq, (uint64_t)a_samp, a_bm, a_peak, u_frames, u_sil, u_clip);
}

// ============================================================================
// SURVEY 8(f) rank 1 — ED-137 RTP depayload + gather (transport_rtp_cb's header parse and payload
// copy, TransportAdapter.cpp:240-292, batched).  One lane per 16-byte piece of the DENSE output:
// writes are full coalesced dwordx4; reads come from pkt + header (4-byte aligned: header 12 or 20,
// slot stride % 4 == 0) as four dwords.  Lane q == 0 of a frame also parses the header and emits
// len / info.  Needs n % 16 == 0; other n use the byte kernel below.
// ============================================================================
struct FrameHdr { uint32_t len; igdsp_rtp_info info; };

__device__ __forceinline__ FrameHdr parse_rtp_words(uint32_t w0, uint32_t w3, uint32_t w4, uint32_t size, uint32_t hdr, bool radio, uint32_t n)
{
    // w0 = packet bytes 0-3, w3 = bytes 12-15 (extension profile / length), w4 = bytes 16-19 (ED-137 word); w3 / w4 are
    // only looked at for radio packets of at least 20 bytes
    FrameHdr r;
    r.len = 0; r.info.ed137 = 0; r.info.payload_len = 0; r.info.pt = 0; r.info.flags = 0;
    if (size < hdr) {
        r.info.flags = IGDSP_RTP_RUNT;
        if (size >= 2u) r.info.pt = (uint8_t)((w0 >> 8) & 0x7Fu);
        return r;
    }
    const uint32_t pt = (w0 >> 8) & 0x7Fu;
    uint32_t fl = (((w0 >> 6) & 3u) == 2u ? IGDSP_RTP_V2 : 0u) | ((w0 & 0x10u) ? IGDSP_RTP_X : 0u) | ((w0 & 0x8000u) ? IGDSP_RTP_MARKER : 0u);
    if (radio) {
        if (pt == 8u || pt == 0u || pt == 18u || pt == 123u) r.info.ed137 = __builtin_bswap32(w4);   // ntohl
        if ((w0 & 0x10u) && w3 == 0x01006701u) fl |= IGDSP_RTP_ED137_OK;                               // bytes 01 67 00 01
    }
    if (pt == 123u) fl |= IGDSP_RTP_KEEPALIVE;
    const uint32_t pl = size - hdr;
    if (pl > n) fl |= IGDSP_RTP_OVERSIZE;
    else if ((pt == 0u || pt == 8u) && pl > 0u) { fl |= IGDSP_RTP_METERED; r.len = pl; }
    r.info.pt = (uint8_t)pt; r.info.payload_len = (uint16_t)pl; r.info.flags = (uint8_t)fl;
    return r;
}

__device__ __forceinline__ FrameHdr parse_rtp(const uint8_t *pkt, uint32_t size, uint32_t hdr, bool radio, uint32_t n)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(pkt);
    const bool wide = radio && size >= hdr;
    return parse_rtp_words(w[0], wide ? w[3] : 0u, wide ? w[4] : 0u, size, hdr, radio, n);
}

// Tuned n == 160 form: a wave takes 64 consecutive packets.  Twelve dword-aligned 16-byte pieces per packet (bytes
// [0,16) and [4,20) of the header, then the ten payload pieces at hdr + 16 k) are spread over the lanes exactly as in
// k_meter_rtp64, so every load instruction covers ~5 whole packets.  The header pieces hand {bytes 0-3, ext word,
// ED-137 word} to the packet's frame lane through LDS; the frame lane parses once per packet (instead of once per
// piece), writes len / info coalesced and publishes the payload length; each payload piece then masks and stores
// itself into the dense output (one contiguous run per store instruction).
constexpr int kDpWaves = 4;
__global__ __launch_bounds__(kDpWaves * 64, 3) void k_depayload64(const uint8_t *__restrict__ packets, const uint16_t *__restrict__ sizes,
                                                               const uint8_t *__restrict__ radio, uint32_t C, uint32_t n_frames,
                                                               uint32_t stride, uint8_t *__restrict__ payload,
                                                               uint16_t *__restrict__ len, igdsp_rtp_info *__restrict__ info)
{
    __shared__ uint4 hdrs[kDpWaves][64];          // per packet {bytes 0-3, -, ext word, ED-137 word}; .y reused for the parsed length
    __shared__ uint4 xp[kDpWaves][kSuperFrames * kPiecesPerFrame];   // 10 KiB per wave: the dense output block, for whole-line stores
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *hw = hdrs[wave];
    const uint32_t n_super = n_frames / kSuperFrames;
    const uint32_t total_waves = gridDim.x * kDpWaves;
    for (uint32_t it = blockIdx.x * kDpWaves + wave; it < n_super; it += total_waves) {
        // items are visited alternately from the two halves of the launch, so that the dense output's two halves — which
        // igdsp_io_alloc puts into two memory classes — are both being written at any moment (as spread_batch does for the
        // persistent kernels)
        const uint32_t sidx = spread_batch(it, n_super);
        // piece p = j * 64 + lane of the item -> packet fr = p / 12, piece q = p % 12.  Recomputed from an opaque copy of
        // the lane id in each phase: hoisting all 12 x 3 sets of lane constants out of the loop costs more registers
        // (and spills) than the few VALU ops they take.
        uint32_t ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t f0 = sidx * kSuperFrames, fi = f0 + lane;
        const bool my_radio = radio[fi % C] != 0;
        const uint32_t my_size = min(sizes ? (uint32_t)sizes[fi] : stride, stride);
        const uint64_t rmask = __ballot(my_radio);
        const uint8_t *base = packets + (uint64_t)f0 * stride;                       // wave-uniform bases, 32-bit lane offsets
        uint4 *ob = reinterpret_cast<uint4 *>(payload) + (uint64_t)f0 * kPiecesPerFrame;
        uint4 d[kSlotPieces];
#pragma unroll
        for (int j = 0; j < kSlotPieces; ++j) {
            const uint32_t p = (uint32_t)j * 64u + ln, fr = p / 12u, q = p - fr * 12u;
            const uint32_t hb = (uint32_t)((rmask >> fr) & 1ull);
            const uint32_t po = q == 0u ? 0u : (q == 1u ? 4u : 12u + 8u * hb + 16u * (q - 2u));
            d[j] = ld16_dw(base + (fr * stride + po));
        }
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < kSlotPieces; ++j) {
            const uint32_t p = (uint32_t)j * 64u + ln, fr = p / 12u, q = p - fr * 12u;
            if (q < 2u)                              // piece 0 -> {.x = bytes 0-3, .y = 0}; piece 1 -> {.z = ext word, .w = ED-137 word}
                reinterpret_cast<uint2 *>(&hw[fr])[q] = q == 0u ? make_uint2(d[j].x, 0u) : make_uint2(d[j].z, d[j].w);
        }
        wave_lds_fence();
        {
            const uint4 h = hw[lane];
            const FrameHdr r = parse_rtp_words(h.x, h.z, h.w, my_size, my_radio ? 20u : 12u, my_radio, (uint32_t)kFrame);
            len[fi] = (uint16_t)r.len;
            info[fi] = r.info;
            hw[lane].y = r.len;
        }
        wave_lds_fence();
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < kSlotPieces; ++j) {
            const uint32_t p = (uint32_t)j * 64u + ln, fr = p / 12u, q = p - fr * 12u;
            if (q >= 2u) {
                const uint32_t flen = hw[fr].y, b0 = 16u * (q - 2u);
                const uint32_t nb = flen > b0 ? min(flen - b0, 16u) : 0u;
                uint32_t x[4] = {d[j].x, d[j].y, d[j].z, d[j].w};
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {  // keep the first nb bytes of the piece, branch-free
                    const uint32_t bits = 8u * (nb > 4u * k ? min(nb - 4u * k, 4u) : 0u);
                    x[k] &= (uint32_t)((1ull << bits) - 1ull);
                }
                xp[wave][fr * (uint32_t)kPiecesPerFrame + (q - 2u)] = make_uint4(x[0], x[1], x[2], x[3]);
            }
        }
        wave_lds_fence();
        // the block is complete in LDS in output order: ten stores of 1 KiB of whole lines each
#pragma unroll
        for (int j = 0; j < kPiecesPerFrame; ++j) ob[(uint32_t)j * 64u + lane] = xp[wave][(uint32_t)j * 64u + lane];
        wave_lds_fence();
    }
}

__global__ __launch_bounds__(256) void k_depayload16(const uint8_t *__restrict__ packets, const uint16_t *__restrict__ sizes,
                                                     const uint8_t *__restrict__ radio, uint32_t C, uint32_t n_frames,
                                                     uint32_t stride, uint32_t n, uint8_t *__restrict__ payload,
                                                     uint16_t *__restrict__ len, igdsp_rtp_info *__restrict__ info)
{
    const uint32_t ppf = n >> 4;                                   // pieces per frame
    const uint64_t n_pieces = (uint64_t)n_frames * ppf;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pieces; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t fi = (uint32_t)(p / ppf), q = (uint32_t)(p - (uint64_t)fi * ppf);
        const uint32_t c = fi % C;
        const bool rad = radio[c] != 0;
        const uint32_t hdr = rad ? 20u : 12u;
        const uint8_t *pkt = packets + (uint64_t)fi * stride;
        const uint32_t size = min(sizes ? (uint32_t)sizes[fi] : stride, stride);
        const FrameHdr h = parse_rtp(pkt, size, hdr, rad, n);
        if (q == 0u) { len[fi] = (uint16_t)h.len; info[fi] = h.info; }
        const uint32_t b0 = q * 16u;
        uint32_t v[4] = {0u, 0u, 0u, 0u};
        if (h.len > b0) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(pkt + hdr + b0);
            const uint32_t nb = min(h.len - b0, 16u);
            if (hdr + b0 + 16u <= stride) {
                // whole piece inside the slot: ONE 16-byte load at dword alignment (gfx950 global loads need only
                // dword alignment for dwordx4) instead of four dword loads, then mask what lies past the length
                struct __attribute__((packed, aligned(4))) Q { uint32_t a, b, c, d; };
                const Q qv = *reinterpret_cast<const Q *>(src);
                const uint32_t x[4] = {qv.a, qv.b, qv.c, qv.d};
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t keep = nb > 4u * k ? min(nb - 4u * k, 4u) : 0u;
                    v[k] = keep == 4u ? x[k] : (keep == 0u ? 0u : (x[k] & ((1u << (8u * keep)) - 1u)));
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k)
                    if (nb > 4u * k) {                              // the dword may extend past the packet's size but never past its slot
                        uint32_t x = (hdr + b0 + 4u * k + 4u <= stride) ? src[k] : 0u;
                        const uint32_t keep = nb - 4u * k;
                        if (keep < 4u) x &= (1u << (8u * keep)) - 1u;
                        v[k] = x;
                    }
            }
        }
        reinterpret_cast<uint4 *>(payload)[p] = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void k_depayload_bytes(const uint8_t *__restrict__ packets, const uint16_t *__restrict__ sizes,
                                                         const uint8_t *__restrict__ radio, uint32_t C, uint32_t n_frames,
                                                         uint32_t stride, uint32_t n, uint8_t *__restrict__ payload,
                                                         uint16_t *__restrict__ len, igdsp_rtp_info *__restrict__ info)
{
    // one wavefront per frame, any n and any (4-byte aligned) stride
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t fi = blockIdx.x * 4u + wave; fi < n_frames; fi += gridDim.x * 4u) {
        const uint32_t c = fi % C;
        const bool rad = radio[c] != 0;
        const uint32_t hdr = rad ? 20u : 12u;
        const uint8_t *pkt = packets + (uint64_t)fi * stride;
        const uint32_t size = min(sizes ? (uint32_t)sizes[fi] : stride, stride);
        const FrameHdr h = parse_rtp(pkt, size, hdr, rad, n);
        if (lane == 0u) { len[fi] = (uint16_t)h.len; info[fi] = h.info; }
        for (uint32_t i = lane; i < n; i += 64u) payload[(uint64_t)fi * n + i] = (i < h.len) ? pkt[hdr + i] : (uint8_t)0;
    }
}

hipError_t launch_decode_meter_rtp(const LaunchCfg &cfg, const uint8_t *slots, const uint16_t *sizes, const uint8_t *codec, uint32_t C,
                                   uint32_t F, uint32_t stride, uint32_t hdr, igdsp_frame_stats *stats, igdsp_rtp_info *info,
                                   igdsp_aggregate *agg, uint32_t rank, hipStream_t s, const uint8_t *radio, const WinArgs *win)
{
    // stride == 0: the 192-byte slot format; otherwise packets packed at `stride` with a `hdr`-byte RTP header, or, with
    // `radio`, a per-channel 20 / 12-byte header.  win: the gated window in the same pass (C % 64 == 0; launch_window_fused)
    const uint32_t n_frames = C * F;                       // caller guarantees a multiple of 64
    if (n_frames == 0) return hipSuccess;
    dim3 blk(kRtpWaves * 64);
    if (win != nullptr) {
        uint32_t waves = kRtpWaves;
        if (const char *e = std::getenv("IGDSP_WIN_WAVES")) waves = (uint32_t)std::max(1, std::min((int)kRtpWaves, std::atoi(e)));   // experiments
        blk = dim3(waves * 64);
        const uint32_t grid = blocks_for((uint64_t)win->n_groups * win->n_seg, waves, (uint32_t)cfg.compute_units);
        uint32_t *noq = nullptr;                           // units are assigned statically (a grid apart), no device queue
        if (win->gpb != 0u) {                              // block-owned channel groups: one block per gpb groups
            if (waves < win->gpb) blk = dim3(win->gpb * 64);      // (a wave per group folds it at the block's end)
            const uint32_t gridb = win->n_groups / win->gpb;
            if (stride == 0)          hipLaunchKernelGGL((k_meter_rtp64<true, true, false, 2>), dim3(gridb), blk, 0, s, slots, sizes, codec, C, n_frames, 192u, 20u, stats, info, agg, rank, noq, radio, *win);
            else if (radio != nullptr) hipLaunchKernelGGL((k_meter_rtp64<true, false, true, 2>), dim3(gridb), blk, 0, s, slots, sizes, codec, C, n_frames, stride, 12u, stats, info, agg, rank, noq, radio, *win);
            else                       hipLaunchKernelGGL((k_meter_rtp64<true, false, false, 2>), dim3(gridb), blk, 0, s, slots, sizes, codec, C, n_frames, stride, hdr, stats, info, agg, rank, noq, radio, *win);
            return hipGetLastError();
        }
        // (one instantiation per layout: without an aggregate the AGG code still runs and wave_exit drops the totals — the
        // aggregate-free packed instantiation needed 170 VGPRs and spilled)
        if (stride == 0)          hipLaunchKernelGGL((k_meter_rtp64<true, true, false, 1>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, 192u, 20u, stats, info, agg, rank, noq, radio, *win);
        else if (radio != nullptr) hipLaunchKernelGGL((k_meter_rtp64<true, false, true, 1>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, stride, 12u, stats, info, agg, rank, noq, radio, *win);
        else                       hipLaunchKernelGGL((k_meter_rtp64<true, false, false, 1>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, stride, hdr, stats, info, agg, rank, noq, radio, *win);
        return hipGetLastError();
    }
    const uint32_t grid = blocks_for(n_frames / kSuperFrames, kRtpWaves, (uint32_t)cfg.compute_units);
    if (stride == 0) {
        if (agg) hipLaunchKernelGGL((k_meter_rtp64<true, true>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, 192u, 20u, stats, info, agg, rank, cfg.gqueue, radio, WinArgs{});
        else     hipLaunchKernelGGL((k_meter_rtp64<false, true>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, 192u, 20u, stats, info, agg, rank, cfg.gqueue, radio, WinArgs{});
    } else if (radio != nullptr) {
        if (agg) hipLaunchKernelGGL((k_meter_rtp64<true, false, true>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, stride, 12u, stats, info, agg, rank, cfg.gqueue, radio, WinArgs{});
        else     hipLaunchKernelGGL((k_meter_rtp64<false, false, true>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, stride, 12u, stats, info, agg, rank, cfg.gqueue, radio, WinArgs{});
    } else {
        if (agg) hipLaunchKernelGGL((k_meter_rtp64<true, false>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, stride, hdr, stats, info, agg, rank, cfg.gqueue, radio, WinArgs{});
        else     hipLaunchKernelGGL((k_meter_rtp64<false, false>), dim3(grid), blk, 0, s, slots, sizes, codec, C, n_frames, stride, hdr, stats, info, agg, rank, cfg.gqueue, radio, WinArgs{});
    }
    return hipGetLastError();
}

hipError_t launch_depayload(const LaunchCfg &cfg, const uint8_t *packets, const uint16_t *sizes, const uint8_t *radio,
                            uint32_t C, uint32_t F, uint32_t stride, uint32_t n, uint8_t *payload, uint16_t *len,
                            igdsp_rtp_info *info, hipStream_t s)
{
    const uint32_t n_frames = C * F;
    if (n_frames == 0) return hipSuccess;
    const uint32_t cap = (uint32_t)cfg.compute_units * 8u;
    const bool aligned = (reinterpret_cast<uintptr_t>(payload) & 15u) == 0u;
    if (aligned && n == (uint32_t)kFrame && (n_frames % kSuperFrames) == 0u && stride >= 180u) {
        // whole super-chunks of n == 160 packets whose slots hold a full radio packet: header parsed once per packet
        hipLaunchKernelGGL(k_depayload64, dim3(blocks_for(n_frames / kSuperFrames, kDpWaves, cap)), dim3(kDpWaves * 64), 0, s,
                           packets, sizes, radio, C, n_frames, stride, payload, len, info);
    } else if ((n & 15u) == 0u && aligned) {
        const uint64_t pieces = (uint64_t)n_frames * (n >> 4);
        hipLaunchKernelGGL(k_depayload16, dim3(blocks_for(pieces, 256, cap)), dim3(256), 0, s, packets, sizes, radio, C, n_frames, stride, n, payload, len, info);
    } else {
        hipLaunchKernelGGL(k_depayload_bytes, dim3(blocks_for(n_frames, 4, cap)), dim3(256), 0, s, packets, sizes, radio, C, n_frames, stride, n, payload, len, info);
    }
    return hipGetLastError();
}


}  // namespace igdsp
