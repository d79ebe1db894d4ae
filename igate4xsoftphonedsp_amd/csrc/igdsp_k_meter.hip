// igdsp_k_meter.hip — decode + meter: k_meter_chunk64 (headline), k_meter_strided, k_meter_image, k_meter_wave_per_frame, k_meter_fat; launch_decode_meter.
// Hand-written gfx950 (CDNA4, wave64) kernels; no MFMA: the path is a byte stream with ~4 integer ops per sample, bounded by
// HBM.  Shared device code: igdsp_device.h.
#include "igdsp_device.h"

namespace igdsp {

// ============================================================================
// Variant 1 — the literal north_star mapping: ONE wavefront per channel-frame.
// Lane l owns bytes [4l, 4l+4) of the frame (n <= 256 => <= 64 lanes; n = 160
// uses 40 lanes), 256-entry int16 expansion LUT per law staged in LDS, wave
// shuffle-reduce.  Handles every n in 1..256, ragged lengths and unaligned
// frames; it is the general fallback of the ABI.
// ============================================================================
__global__ __launch_bounds__(256) void k_meter_wave_per_frame(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, const uint16_t *__restrict__ len,
    uint32_t C, uint32_t first_frame, uint32_t n_frames, uint32_t n, igdsp_frame_stats *__restrict__ stats,
    int16_t *__restrict__ pcm, igdsp_aggregate *agg, uint32_t rank)
{
    // frames [first_frame, n_frames) of the batch; all pointers are the batch bases
    __shared__ int16_t lut[2][256];
    __shared__ uint2 agg_slots[4 * 4];
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) {
        const uint32_t code = i & 255u;
        const int ax = (int)((i >> 8) ? alaw_abs(code) : ulaw_abs(code));
        lut[i >> 8][code] = (int16_t)((code & 0x80u) ? ax : -ax);
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool dword_ok = ((n & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(payload) & 3u) == 0u) &&
                          ((reinterpret_cast<uintptr_t>(pcm) & 7u) == 0u);
    uint64_t a_sumsq = 0, a_samples = 0;
    uint32_t a_frames = 0, a_sil = 0, a_clip = 0, a_bm = 0, a_peak = 0;

    auto load_frame = [&](uint32_t fi) -> uint32_t {                 // this lane's four payload bytes of frame fi
        const uint8_t *base = payload + (uint64_t)fi * n;
        const uint32_t b0 = lane * 4u;
        uint32_t w = 0;
        if (dword_ok) {
            if (b0 < n) w = *reinterpret_cast<const uint32_t *>(base + b0);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
                if (b0 + k < n) w |= (uint32_t)base[b0 + k] << (8u * k);
        }
        return w;
    };
    auto process_frame = [&](uint32_t fi, uint32_t w) {
        const uint32_t c = fi % C;
        const bool alaw = codec[c] == IGDSP_PT_PCMA;
        uint32_t l = len ? (uint32_t)len[fi] : n;
        l = min(l, n);
        const uint32_t b0 = lane * 4u;
        const uint32_t nvalid = (l > b0) ? min(l - b0, 4u) : 0u;
        uint32_t sum = 0, peak = 0, bsum = 0;   // sum of (|x|/4)^2: 4 * 8064^2 = 2.6e8 per lane
        int x[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t b = (w >> (8u * k)) & 255u;
            int v = lut[alaw][b];
            if (k >= nvalid) v = 0;
            x[k] = v;
            const uint32_t ax = (uint32_t)(v < 0 ? -v : v);
            sum += (ax >> 2) * (ax >> 2);        // every G.711 magnitude is a multiple of 4
            peak = max(peak, ax);
            bsum += (k < nvalid) ? b : 0u;
        }
        if (pcm != nullptr && b0 < n) {
            int16_t *o = pcm + (uint64_t)fi * n + b0;
            if (dword_ok) {
                uint2 pk;
                pk.x = ((uint32_t)x[0] & 0xFFFFu) | ((uint32_t)x[1] << 16);
                pk.y = ((uint32_t)x[2] & 0xFFFFu) | ((uint32_t)x[3] << 16);
                *reinterpret_cast<uint2 *>(o) = pk;
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k)
                    if (b0 + k < n) o[k] = (int16_t)x[k];
            }
        }
        // reference silence probe: payload bytes 28 / 38 / 48 (lanes 7, 9, 12)
        const uint32_t w7 = (uint32_t)__builtin_amdgcn_readlane((int)w, 7), w9 = (uint32_t)__builtin_amdgcn_readlane((int)w, 9),
                       w12 = (uint32_t)__builtin_amdgcn_readlane((int)w, 12);
        const bool probe = (l > 48u) && ((w7 & 255u) == 0xD5u) && (((w9 >> 16) & 255u) == 0xD5u) && ((w12 & 255u) == 0xD5u);
        // wavefront shuffle-reduce: the 38-bit sum travels as two 32-bit halves (low 16 bits / rest)
        const uint32_t r_lo = wave_reduce_dpp(sum & 0xFFFFu, OpAdd()), r_hi = wave_reduce_dpp(sum >> 16, OpAdd());
        const uint64_t s64 = (((uint64_t)r_hi << 16) + r_lo) << 4;      // x^2 = 16 * (|x|/4)^2
        peak = wave_reduce_dpp(peak, OpMax());
        bsum = wave_reduce_dpp(bsum, OpAdd());
        if (lane == 0) {
            igdsp_frame_stats st;
            if (l == 0u) {
                st.sumsq = 0; st.rms = 0.f; st.peak = 0; st.byte_mean = 0; st.flags = IGDSP_FLAG_EMPTY;
            } else {
                st = make_stats(s64, peak, bsum, l, alaw, probe);
                a_sumsq += s64; a_samples += l; a_frames += 1; a_sil += (st.flags & IGDSP_FLAG_SILENT) ? 1u : 0u;
                a_clip += (st.flags & IGDSP_FLAG_CLIPPED) ? 1u : 0u; a_bm += st.byte_mean; a_peak = max(a_peak, peak);
            }
            stats[fi] = st;
        }
    };
    // One wavefront per channel-frame, eight frames in flight per wave: a single 160-byte load per wave would
    // leave ~5 KB in flight per CU (0.5 TB/s); the eight loads of consecutive frames are issued back to back
    // (tail indices clamped so no load is conditional) and then folded one frame at a time.
    constexpr uint32_t U = 8;
    const uint32_t last = n_frames - 1u;
    for (uint32_t f0 = first_frame + (blockIdx.x * 4u + wave) * U; f0 < n_frames; f0 += gridDim.x * 4u * U) {
        uint32_t w[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) w[u] = load_frame(min(f0 + u, last));
#pragma unroll
        for (uint32_t u = 0; u < U; ++u)
            if (f0 + u < n_frames) process_frame(f0 + u, w[u]);     // wave-uniform condition
    }
    if (agg != nullptr) agg_commit_block(agg, rank, agg_slots, 4u, a_sumsq, a_samples, a_frames, a_sil, a_clip, a_bm, a_peak);
}

// ----------------------------------------------------------------------------
// One half (32 frames = five 16-byte pieces per lane) of a super-chunk: expand, square-accumulate,
// peak, byte-sum, probe; one strip entry per piece.  The 80 LUT reads are software-pipelined in
// units of 8 samples: unit u+1's eight ds_read_b64 are in flight while unit u is folded, so a wave
// hides most LDS latency by itself (at most 16 LDS reads outstanding = the lgkmcnt limit).
// ----------------------------------------------------------------------------
template <bool STORE_PCM>
__device__ __forceinline__ void process_half(const uint2 *lut, uint2 *strip_half, uint4 (&d)[kLoadsPerChunk],
                                             const uint32_t am, const uint32_t (&fr)[kLoadsPerChunk], const uint32_t (&pm)[kLoadsPerChunk],
                                             const uint32_t off, const uint32_t lane, uint4 *pcm_half, const uint4 *refill,
                                             uint4 *xpose = nullptr)
{
    // `am` = the A-law ballot of this half's 32 frames (frame l of the half in bit l); a piece picks its frame's bit with
    // one v_bfe_i32 when it is expanded, so no per-piece law-mask array stays live across the half (register pressure).
    // `refill` = this lane's first piece of the NEXT super-chunk's same half: piece j's register is
    // reloaded the moment piece j has been folded, so the five loads trickle out evenly and get
    // most of an iteration of lead time.
    uint2 e[2][8];
    uint32_t wa[2], wb[2];
    auto issue = [&](int u) {
        const int j = u >> 1, k = u & 1;
        wa[k] = (u & 1) ? d[j].z : d[j].x;
        wb[k] = (u & 1) ? d[j].w : d[j].y;
        const uint32_t lmj = (uint32_t)__builtin_amdgcn_sbfe(am, fr[j], 1) & 0x80808080u;
        const uint32_t ta = (wa[k] & 0x7F7F7F7Fu) | lmj, tb = (wb[k] & 0x7F7F7F7Fu) | lmj;
        e[k][0] = lut_at(lut, ta, off, 0x0C0C0400u); e[k][1] = lut_at(lut, ta, off, 0x0C0C0500u);
        e[k][2] = lut_at(lut, ta, off, 0x0C0C0600u); e[k][3] = lut_at(lut, ta, off, 0x0C0C0700u);
        e[k][4] = lut_at(lut, tb, off, 0x0C0C0400u); e[k][5] = lut_at(lut, tb, off, 0x0C0C0500u);
        e[k][6] = lut_at(lut, tb, off, 0x0C0C0600u); e[k][7] = lut_at(lut, tb, off, 0x0C0C0700u);
    };
    uint32_t sum = 0, peak = 0, bsum = 0;
    uint32_t o[8];
    issue(0);
#pragma unroll
    for (int u = 0; u < 2 * kLoadsPerChunk; ++u) {
        const int j = u >> 1, k = u & 1;
        if (u + 1 < 2 * kLoadsPerChunk) issue(u + 1);
        __builtin_amdgcn_sched_barrier(0);      // keep the next unit's reads ahead of this unit's folds
        bsum = __builtin_amdgcn_sad_u8(wa[k], 0u, bsum);
        bsum = __builtin_amdgcn_sad_u8(wb[k], 0u, bsum);
        sum = sum + e[k][0].x + e[k][1].x; sum = sum + e[k][2].x + e[k][3].x;
        sum = sum + e[k][4].x + e[k][5].x; sum = sum + e[k][6].x + e[k][7].x;
        peak = max(max(peak, e[k][0].y), e[k][1].y); peak = max(max(peak, e[k][2].y), e[k][3].y);
        peak = max(max(peak, e[k][4].y), e[k][5].y); peak = max(max(peak, e[k][6].y), e[k][7].y);
        if (STORE_PCM) {
            o[4 * k + 0] = pack_pcm(wa[k], 0, e[k][0].y, e[k][1].y); o[4 * k + 1] = pack_pcm(wa[k], 2, e[k][2].y, e[k][3].y);
            o[4 * k + 2] = pack_pcm(wb[k], 0, e[k][4].y, e[k][5].y); o[4 * k + 3] = pack_pcm(wb[k], 2, e[k][6].y, e[k][7].y);
        }
        if (k == 1) {                           // piece j complete
            strip_half[j * 64 + lane] = make_uint2(sum, peak | (bsum << 16) | probe_fail(d[j], pm[j]));
            if (STORE_PCM) {
                // Each lane holds 32 contiguous PCM bytes (A = o[0..3], B = o[4..7]): stored from the lanes' own registers they
                // would leave as 16-byte pieces at 32-byte stride.  (History, DESIGN 3.4b: a quad-local DPP regroup wrote 64
                // contiguous bytes per quad; nontemporal stores measured 18 % slower on this pattern.)
                // transposition through a per-wave 2 KiB LDS scratch: lane l parks its 32 bytes at l * 32, then reads back
                // bytes [16 l, 16 l + 16) of each KiB, so both store instructions write 1 KiB contiguous (whole lines)
                // (16-byte unit u lives at u ^ ((u >> 3) & 1): lanes l and l + 4 of a ds_write_b128 pass would otherwise meet in the
                // same four banks — 2.1e7 conflict cycles per launch, SQ_LDS_BANK_CONFLICT; reads permute inside aligned groups of 8)
                const uint32_t wsw = (lane >> 2) & 1u, rsw = lane ^ ((lane >> 3) & 1u);
                xpose[(2u * lane) ^ wsw] = make_uint4(o[0], o[1], o[2], o[3]);
                xpose[(2u * lane + 1u) ^ wsw] = make_uint4(o[4], o[5], o[6], o[7]);
                wave_lds_fence();
                const uint4 v0 = xpose[rsw], v1 = xpose[64u + rsw];
                wave_lds_fence();
                uint4 *op = pcm_half + ((uint32_t)j * 128u + lane);
                op[0] = v0;
                op[64] = v1;
            }
            d[j] = ld_stream(refill + j * 64);
            sum = 0; peak = 0; bsum = 0;
        }
    }
}

// ============================================================================
// Variant 2 (default for n == 160) — "chunk64".
//
// Work unit: a super-chunk of 64 consecutive channel-frames = 10 240 contiguous bytes, owned by ONE
// wavefront and fetched as ten wave-wide 16 B/lane loads (1 KiB per instruction, fully coalesced).
// A 16-byte piece never straddles a frame (160 = 10 x 16), so each lane reduces its ten pieces
// privately; the 10 pieces of every frame are then folded by that frame's lane (all 64 lanes busy)
// through a per-wave LDS strip, and 64 x 16 B records leave as one 1 KiB store.
//
// Pipeline per wave: registers X / Y hold the two 32-frame halves.  While half X is expanded the
// loads refilling Y (issued half an iteration earlier) are in flight, and vice versa; no load in the
// steady-state loop is conditional (tail pieces are clamped, the final prefetch re-reads the current
// super-chunk) so the compiler keeps counted vmcnt waits.
//
// Balance: block b owns super-chunks b, b+G, b+2G, ...; its 16 waves pull the next one from an LDS
// counter, so the waves of a CU finish within one iteration of each other.
// ============================================================================
// waves per block: 16 (1024 threads, 128 VGPRs) for the meter-only kernel: a read-heavy kernel wants every wave it can get, each
// has one super-chunk of loads in flight.  The PCM-store variant writes two bytes for every byte it reads, and a 1 : 2 mix is
// fastest with FEW resident waves per CU — 12 / 10 / 8 / 6 / 5 / 4 / 3 / 2 waves: 0.6706 / 0.6680 / 0.6636 / 0.6573 / 0.6520 / 0.6388 / 0.6995 /
// 1.013 ms in same-box A/B builds (late round 3; 12 had been chosen for its registers) — every wave is one more front of 20 KiB
// write bursts, and four already keep enough loads in flight.
#ifndef IGDSP_STORE_WAVES
#define IGDSP_STORE_WAVES 4
#endif
template <bool STORE_PCM> struct ChunkGeom { static constexpr int kWaves = STORE_PCM ? IGDSP_STORE_WAVES : kWavesPerBlock; };

// DIAG: a separate diagnostic instantiation (never the shipped path) that stamps where a
// wave's cycles go; the stamps leave only through `diag`, no output is computed from them.
template <bool STORE_PCM, bool AGG, bool DIAG = false>
__global__ __launch_bounds__(ChunkGeom<STORE_PCM>::kWaves * 64) void k_meter_chunk64(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t n_frames,
    igdsp_frame_stats *__restrict__ stats, int16_t *__restrict__ pcm, igdsp_aggregate *agg, uint32_t rank,
    uint64_t *__restrict__ diag = nullptr, uint32_t *__restrict__ gqueue = nullptr)
{
    constexpr int kWaves = ChunkGeom<STORE_PCM>::kWaves;
    __shared__ uint2 lds[kLutEntries + kWaves * kStripEntries + (STORE_PCM ? kWaves * 256 : 0)];        // LUT + 5 KiB strip per wave (+ 2 KiB PCM transposition scratch)
    // Work queue.  A *batch* = kWaves consecutive super-chunks.  The block's first batch is static (its
    // blockIdx); further batches come from ONE device-wide counter (gqueue[0], one atomic per batch, i.e.
    // per ~160 KiB of input), so fast CUs take more and the launch has no inter-CU tail.  Inside the block
    // the waves draw slots from an LDS counter; the wave that draws the first slot of local batch j
    // prefetches the id of batch j+1, so nobody waits for the device atomic's latency.
    constexpr int kRing = 8;
    __shared__ uint32_t q_next, q_batch[kRing], q_tag[kRing];
    __shared__ AggBlock aggb;
    uint64_t d_t0 = 0, d_t1 = 0, d_iter = 0, d_rt0 = 0, d_setup = 0, d_px = 0, d_py = 0, d_red = 0;
    if (DIAG) { d_t0 = now_cycles(); d_rt0 = __builtin_amdgcn_s_memrealtime(); }
    const uint32_t G = gridDim.x;
    uint32_t gb1 = 0;
    if (threadIdx.x == 0 && gqueue != nullptr) gb1 = atomicAdd(gqueue, 1u);   // id of this block's 2nd batch; lands under the LUT fill
    fill_lut(lds);
    if (threadIdx.x == 0) {
        q_next = kWaves;                                         // slots 0..kWaves-1 = the waves' first picks
        agg_block_init(aggb);
        for (int i = 0; i < kRing; ++i) q_tag[i] = 0xFFFFFFFFu;
        q_batch[0] = blockIdx.x; q_tag[0] = 0u;
        q_batch[1] = gqueue ? gb1 + G : blockIdx.x + G; q_tag[1] = 1u;
    }
    __syncthreads();
    if (DIAG) d_t1 = now_cycles();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint2 *strip = lds + kLutEntries + wave * kStripEntries;
    uint4 *xpose = STORE_PCM ? reinterpret_cast<uint4 *>(lds + kLutEntries + kWaves * kStripEntries + wave * 256) : nullptr;
    const uint32_t off = (lane & 31u) * 8u;

    uint32_t fr[kLoadsPerChunk], pm[kLoadsPerChunk];   // frame-in-half / probe mask of this lane's five pieces per half
#pragma unroll
    for (int j = 0; j < kLoadsPerChunk; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane;
        fr[j] = p / 10u;
        pm[j] = probe_mask(p - fr[j] * 10u);
    }

    // the two halves are only visited alternately when there is a bulk output to spread (PCM); for the meter alone the
    // plain ascending order is as fast on average and steadier from launch to launch
    const uint32_t n_batches = (STORE_PCM || IGDSP_SPREAD_METER) ? (n_frames / kSuperFrames + (uint32_t)kWaves - 1u) / (uint32_t)kWaves : 0u;
    const uint32_t n_super = n_frames / kSuperFrames;             // the launcher hands over whole super-chunks only:
    const uint4 *src16 = reinterpret_cast<const uint4 *>(payload); // no tail predicate anywhere in the loop

    // launch-aggregate partials.  Per-lane (VGPR): sum of squares, byte-mean sum, peak.  The three COUNTS are wave-uniform
    // (every iteration meters 64 frames; silent / clipped come from a ballot + popcount) and live in SGPRs, which keeps
    // the meter-only kernel inside its 128-VGPR budget without scratch.
    uint64_t a_sumsq = 0;
    uint32_t a_bm = 0, a_peak = 0;
    uint32_t u_frames = 0, u_sil = 0, u_clip = 0;

    auto fetch_half = [&](uint4 (&dst)[kLoadsPerChunk], uint32_t sidx, uint32_t half) {
        const uint4 *p0 = src16 + ((uint64_t)sidx * (uint32_t)kStripEntries + half * (uint32_t)kPiecesPerChunk + lane);   // 64-bit piece index
#pragma unroll
        for (int j = 0; j < kLoadsPerChunk; ++j) dst[j] = ld_stream(p0 + j * 64);
    };
    auto fetch_pt = [&](uint32_t sidx) {                         // codec id (RTP PT) of this lane's own frame
        const uint32_t c = (sidx * (uint32_t)kSuperFrames + lane) % C;     // < 2^32: the ABI caps C*F
        return (uint32_t)codec[c];
    };
    auto grab = [&]() -> uint32_t {                              // next super-chunk for this wave (wave-uniform)
        uint32_t v = 0;
        if (lane == 0) {
            const uint32_t s = atomicAdd(&q_next, 1u);
            const uint32_t j = s / (uint32_t)kWaves, w = s - j * (uint32_t)kWaves;
            if (w == 0u) {                                       // first drawer of local batch j announces batch j + 1
                const uint32_t nb = gqueue ? atomicAdd(gqueue, 1u) + G : blockIdx.x + (j + 1u) * G;
                __hip_atomic_store(&q_batch[(j + 1u) % kRing], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&q_tag[(j + 1u) % kRing], j + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            while (__hip_atomic_load(&q_tag[j % kRing], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != j)
                __builtin_amdgcn_s_sleep(2);                     // published by a wave of this block that never waits on us
            v = spread_batch(__hip_atomic_load(&q_batch[j % kRing], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), n_batches) * (uint32_t)kWaves + w;
        }
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };

    uint32_t sidx = spread_batch(blockIdx.x, n_batches) * (uint32_t)kWaves + wave;   // batch blockIdx.x, slot = wave
    if (sidx < n_super) {
        uint4 X[kLoadsPerChunk], Y[kLoadsPerChunk];
        uint32_t cur_pt = fetch_pt(sidx);          // issue order pt, X, Y — the same in the prologue and in the loop,
        fetch_half(X, sidx, 0);                    // so the waits at the loop head stay counted (vmcnt(N), not 0)
        fetch_half(Y, sidx, 1);
        uint32_t s_next = grab();                  // the item after this one (pulled one iteration ahead of use)
        for (;;) {
            uint64_t d_a = 0, d_b = 0, d_c = 0, d_d = 0;
            if (DIAG) d_a = now_cycles();
            const bool has_next = s_next < n_super;
            const uint32_t s_load = has_next ? s_next : 0u;      // last round: every wave re-reads super-chunk 0 (L2-hot), loads stay unconditional
            const uint32_t f0 = sidx * kSuperFrames;
            // law of frame l of this super-chunk lives in lane l; one ballot turns it into a 64-bit wave mask,
            // and each piece picks its frame's bit (no cross-lane traffic per piece)
            const bool my_alaw = cur_pt == IGDSP_PT_PCMA;
            const uint64_t amask = __ballot(my_alaw);
            const uint32_t am_lo = (uint32_t)amask, am_hi = (uint32_t)(amask >> 32);
            uint4 *pcm16 = STORE_PCM ? reinterpret_cast<uint4 *>(pcm + (uint64_t)f0 * kFrame) : nullptr;
            const uint4 *nsrc = src16 + ((uint64_t)s_load * (uint32_t)kStripEntries + lane);   // 64-bit: 10 pieces per frame pass 2^32 at 68.7 GB
            const uint32_t nxt_pt = fetch_pt(s_load);
            if (DIAG) d_b = now_cycles();
            process_half<STORE_PCM>(lds, strip, X, am_lo, fr, pm, off, lane, pcm16, nsrc, xpose);
            if (DIAG) d_c = now_cycles();
            process_half<STORE_PCM>(lds, strip + kPiecesPerChunk, Y, am_hi, fr, pm, off, lane, pcm16 + 2 * kPiecesPerChunk, nsrc + kPiecesPerChunk, xpose);
            if (DIAG) d_d = now_cycles();
            const uint32_t s_after = has_next ? grab() : 0xFFFFFFFFu;   // its LDS round trip hides under the frame fold below

            wave_lds_fence();
            {
                const uint4 *row = reinterpret_cast<const uint4 *>(strip + lane * kPiecesPerFrame);   // 80 B rows, 16 B aligned
                uint64_t s = 0;
                uint32_t peak = 0, bsum = 0, fail = 0;
#pragma unroll
                for (int i = 0; i < kPiecesPerFrame / 2; ++i) {
                    const uint4 v = row[i];
                    s += (uint64_t)(v.x + v.z);                   // two 30-bit piece sums fit 32 bits
                    peak = max(max(peak, v.y & 0x7FFFu), v.w & 0x7FFFu);
                    bsum += ((v.y >> 16) & 0x7FFFu) + ((v.w >> 16) & 0x7FFFu);
                    fail |= v.y | v.w;
                }
                uint32_t bm, fl;
                st_stream(reinterpret_cast<uint4 *>(stats + (f0 + lane)), pack_stats160(s, peak, bsum, my_alaw, (fail >> 31) == 0u, bm, fl));
                if (AGG) {
                    a_sumsq += s << 4; a_bm += bm; a_peak = max(a_peak, peak);
                    u_frames += (uint32_t)kSuperFrames;
                    u_sil += (uint32_t)__builtin_popcountll(__ballot((fl & IGDSP_FLAG_SILENT) != 0u));
                    u_clip += (uint32_t)__builtin_popcountll(__ballot((fl & IGDSP_FLAG_CLIPPED) != 0u));
                }
            }
            wave_lds_fence();
            if (DIAG) { d_iter += 1; d_setup += d_b - d_a; d_px += d_c - d_b; d_py += d_d - d_c; d_red += now_cycles() - d_d; }
            if (!has_next) break;
            sidx = s_next;
            s_next = s_after;
            cur_pt = nxt_pt;
        }
    }
    if (DIAG && lane == 0 && diag != nullptr) {
        uint64_t *o = diag + (uint64_t)(blockIdx.x * kWaves + wave) * 12u;
        o[0] = d_t0; o[1] = d_t1; o[2] = now_cycles(); o[3] = d_setup; o[4] = d_px; o[5] = d_iter; o[6] = d_py;
        o[7] = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID, bits [3:0]
        o[8] = d_rt0; o[9] = __builtin_amdgcn_s_memrealtime(); o[10] = d_red; o[11] = wave;
    }
    // End of the wave's work, no barrier: its aggregate partials go into the block accumulator; the block's last wave out
    // commits the block totals and counts the block as finished (the last BLOCK out re-arms the device queue for the next launch).
    wave_exit(AGG ? agg : nullptr, rank, aggb, (uint32_t)kWaves, lane, gqueue, G, a_sumsq, lane == 0u ? (uint64_t)u_frames * kFrame : 0ull,
              a_bm, a_peak, u_frames, u_sil, u_clip);
}

// ============================================================================
// Variant 3 — "fat waves": the same super-chunk pipeline with FOUR super-chunks of lookahead per wave.
// 8 waves/block at up to 256 VGPRs: forty 16-byte piece registers per lane (4 x 10 KiB in flight per
// wave, 320 KiB per CU instead of 160 KiB) — the experiment for the "16 waves/CU recycle their ten
// load registers too slowly" bound of variant 2.  Static interleaved distribution; item k of a wave
// lives in register set k % 4 and each piece is re-loaded from item k + 4 the moment it is folded.
// ============================================================================
constexpr int kFatWaves = 8;
constexpr int kFatDepth = 4;

template <bool AGG>
__global__ __launch_bounds__(kFatWaves * 64) void k_meter_fat(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t n_frames,
    igdsp_frame_stats *__restrict__ stats, igdsp_aggregate *agg, uint32_t rank)
{
    __shared__ uint2 lds[kLutEntries + kFatWaves * kStripEntries];
    fill_lut(lds);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint2 *strip = lds + kLutEntries + wave * kStripEntries;
    const uint32_t off = (lane & 31u) * 8u;
    uint32_t fr[kLoadsPerChunk], pm[kLoadsPerChunk];
#pragma unroll
    for (int j = 0; j < kLoadsPerChunk; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane;
        fr[j] = p / 10u;
        pm[j] = probe_mask(p - fr[j] * 10u);
    }
    const uint32_t n_super = n_frames / kSuperFrames;
    const uint32_t stride = gridDim.x * kFatWaves;                 // items of one wave: first, first + stride, ...
    const uint32_t first = blockIdx.x * kFatWaves + wave;
    const uint4 *src16 = reinterpret_cast<const uint4 *>(payload);

    uint64_t a_sumsq = 0;
    uint32_t a_bm = 0, a_peak = 0;
    uint32_t u_frames = 0, u_sil = 0, u_clip = 0;     // wave-uniform counts (SGPRs), as in k_meter_chunk64

    uint4 X[kFatDepth][kLoadsPerChunk], Y[kFatDepth][kLoadsPerChunk];
    uint32_t pt[kFatDepth];
    auto item_or0 = [&](uint32_t k) { const uint32_t i = first + k * stride; return i < n_super ? i : 0u; };
    auto fetch_pt = [&](uint32_t sidx) { return (uint32_t)codec[(sidx * (uint32_t)kSuperFrames + lane) % C]; };

    if (first < n_super) {
#pragma unroll
        for (int s = 0; s < kFatDepth; ++s) {
            const uint32_t i = item_or0((uint32_t)s);
            pt[s] = fetch_pt(i);
            const uint4 *p0 = src16 + ((uint64_t)i * (uint32_t)kStripEntries + lane);
#pragma unroll
            for (int j = 0; j < kLoadsPerChunk; ++j) X[s][j] = ld_stream(p0 + j * 64);
#pragma unroll
            for (int j = 0; j < kLoadsPerChunk; ++j) Y[s][j] = ld_stream(p0 + kPiecesPerChunk + j * 64);
        }
        for (uint32_t k0 = 0;; k0 += kFatDepth) {
            bool done = false;
#pragma unroll
            for (int s = 0; s < kFatDepth; ++s) {
                const uint32_t sidx = first + (k0 + (uint32_t)s) * stride;
                if (sidx >= n_super) { done = true; break; }
                const uint32_t s_load = item_or0(k0 + (uint32_t)s + kFatDepth);
                const uint32_t f0 = sidx * kSuperFrames;
                const bool my_alaw = pt[s] == IGDSP_PT_PCMA;
                const uint64_t amask = __ballot(my_alaw);
                const uint32_t am_lo = (uint32_t)amask, am_hi = (uint32_t)(amask >> 32);
                const uint4 *nsrc = src16 + ((uint64_t)s_load * (uint32_t)kStripEntries + lane);
                pt[s] = fetch_pt(s_load);
                process_half<false>(lds, strip, X[s], am_lo, fr, pm, off, lane, nullptr, nsrc);
                process_half<false>(lds, strip + kPiecesPerChunk, Y[s], am_hi, fr, pm, off, lane, nullptr, nsrc + kPiecesPerChunk);
                wave_lds_fence();
                {
                    const uint4 *row = reinterpret_cast<const uint4 *>(strip + lane * kPiecesPerFrame);
                    uint64_t sm = 0;
                    uint32_t peak = 0, bsum = 0, fail = 0;
#pragma unroll
                    for (int i = 0; i < kPiecesPerFrame / 2; ++i) {
                        const uint4 v = row[i];
                        sm += (uint64_t)(v.x + v.z);
                        peak = max(max(peak, v.y & 0x7FFFu), v.w & 0x7FFFu);
                        bsum += ((v.y >> 16) & 0x7FFFu) + ((v.w >> 16) & 0x7FFFu);
                        fail |= v.y | v.w;
                    }
                    uint32_t bm, fl;
                    st_stream(reinterpret_cast<uint4 *>(stats + (f0 + lane)), pack_stats160(sm, peak, bsum, my_alaw, (fail >> 31) == 0u, bm, fl));
                    if (AGG) {
                        a_sumsq += sm << 4; a_bm += bm; a_peak = max(a_peak, peak);
                        u_frames += (uint32_t)kSuperFrames;
                        u_sil += (uint32_t)__builtin_popcountll(__ballot((fl & IGDSP_FLAG_SILENT) != 0u));
                        u_clip += (uint32_t)__builtin_popcountll(__ballot((fl & IGDSP_FLAG_CLIPPED) != 0u));
                    }
                }
                wave_lds_fence();
            }
            if (done) break;
        }
    }
    if (AGG && agg != nullptr) {
        const bool l0 = lane == 0u;
        agg_commit_block(agg, rank, lds + kLutEntries, (uint32_t)kFatWaves, a_sumsq, l0 ? (uint64_t)u_frames * kFrame : 0ull, l0 ? u_frames : 0u,
                         l0 ? u_sil : 0u, l0 ? u_clip : 0u, a_bm, a_peak);
    }
}

// ============================================================================
// Every other frame size — k_meter_image: decode + meter for any n with n % 4 == 0 (24, 80, 164, 240 ... the reference's
// hook anticipates 164 and 24, roip_ed137.cpp:6561-6562), optional per-frame lengths, whole super-chunks AND the tail.
//
// A 16-byte piece of a [F][C][n] stream straddles frames when n % 16 != 0, so the piece / strip bookkeeping of
// k_meter_chunk64 does not carry over.  Instead the wave copies its super-chunk (64 frames = 64 n contiguous bytes, fetched
// with the same wave-wide 16 B/lane nontemporal loads) into an LDS image and then lane l meters FRAME l on its own: n / 4
// steps of {one ds_read_b32 of its frame, four LUT reads, accumulate}.  All 64 lanes stay busy for any n, no cross-lane
// fold exists, the probe bytes and the per-frame length are the lane's own, and the 64 records leave as one 1 KiB store.
// The pieces of the NEXT super-chunk are already in flight (in registers) while the current image is metered.
// Lane l starts at dword l * n / 4 of the image: conflict-free when n / 4 is odd (164), 2- to 8-way for the image reads
// (one LDS read in five) when it is even; the LUT reads are conflict-free as everywhere (replica = lane & 31).
// LDS: 64 KiB LUT + 64 n bytes of image per wave, so the block runs min(12, 94 KiB / 64 n) waves (9 at n = 164).
// Algorithmic bytes per sample: (n + 1 + 16) / n.
// ============================================================================
constexpr int kImgMaxPieces = IGDSP_MAX_PAYLOAD * kSuperFrames / 16 / 64;      // 16 wave-wide loads cover 64 frames of 256 bytes

constexpr uint32_t kImgMaxWaves = 12;          // 768 threads: up to 170 VGPRs, room for the sixteen piece registers of the next item

template <bool AGG, bool RAGGED>
__global__ __launch_bounds__(kImgMaxWaves * 64) void k_meter_image(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, const uint16_t *__restrict__ len, uint32_t C,
    uint32_t first_frame, uint32_t n_frames, uint32_t n, igdsp_frame_stats *__restrict__ stats, igdsp_aggregate *agg, uint32_t rank)
{
    // frames [first_frame, n_frames) of the batch in items of 64; all pointers are the batch bases; payload + first_frame * n
    // is dword aligned (launcher)
    __shared__ uint2 lut[kLutEntries];                            // static, at LDS offset 0: LUT addresses need no base add
    extern __shared__ __attribute__((aligned(16))) uint8_t img_smem[];   // the waves' images (sized at launch)
    const uint32_t n_waves = blockDim.x >> 6;
    fill_lut(lut);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t img_bytes = (uint32_t)kSuperFrames * n;
    uint32_t *img = reinterpret_cast<uint32_t *>(img_smem + (size_t)wave * img_bytes);
    const uint32_t off = (lane & 31u) * 8u;
    const uint32_t D = n >> 2;                                   // dwords per frame
    const uint32_t np = (img_bytes / 16u + 63u) >> 6;            // wave-wide loads per item (<= 16)
    const uint32_t n_items = (n_frames - first_frame + (uint32_t)kSuperFrames - 1u) / (uint32_t)kSuperFrames;
    const uint64_t total_bytes = (uint64_t)n_frames * n;
    const uint8_t *base0 = payload + (uint64_t)first_frame * n;

    uint64_t a_sumsq = 0;
    uint32_t a_samp = 0, a_bm = 0, a_peak = 0;
    uint32_t u_frames = 0, u_sil = 0, u_clip = 0;

    uint4 d[kImgMaxPieces];
    // bytes of item `it` that exist in the batch (the last item may be short)
    auto avail_of = [&](uint32_t it) {
        const uint64_t b = (uint64_t)first_frame * n + (uint64_t)it * img_bytes;
        return (uint32_t)min((uint64_t)img_bytes, total_bytes - b);
    };
    // Every load is unconditional (a piece that does not exist in the batch re-reads the batch's first 16 bytes and is never
    // stored): no load result is merged with another value, so the compiler keeps counted vmcnt waits and the pieces of the next
    // item really are in flight while the current image is metered.
    auto fetch = [&](uint32_t it) {
        const uint8_t *src = base0 + (uint64_t)it * img_bytes;
        const uint32_t avail = avail_of(it);
#pragma unroll
        for (int j = 0; j < kImgMaxPieces; ++j)
            if ((uint32_t)j < np) {                              // wave-uniform
                const uint32_t o = ((uint32_t)j * 64u + lane) * 16u;
                const uint8_t *a = (o + 16u <= avail) ? src + o : payload;
                d[j] = ld16_dw(a);                                 // dword alignment is enough (n % 4 == 0, dword-aligned batch)
            }
    };
    const uint32_t stride_items = gridDim.x * n_waves;
    uint32_t item = blockIdx.x * n_waves + wave;
    if (item < n_items) fetch(item);
    for (; item < n_items; item += stride_items) {
        // registers -> image (the previous item's readers are this same wave: program order + fence)
        const uint32_t avail = avail_of(item);
#pragma unroll
        for (int j = 0; j < kImgMaxPieces; ++j)
            if ((uint32_t)j < np) {
                const uint32_t o = ((uint32_t)j * 64u + lane) * 16u;
                if (o + 16u <= avail) *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(img) + o) = d[j];
            }
        if ((avail & 15u) != 0u) {                               // wave-uniform, once per launch at most: the batch ends inside a piece
            const uint32_t o = (avail & ~15u) + 4u * lane;       // its 1..3 dwords, one lane each
            if (o < avail) img[o >> 2] = *reinterpret_cast<const uint32_t *>(base0 + (uint64_t)item * img_bytes + o);
        }
        const uint32_t nxt = item + stride_items;
        if (nxt < n_items) fetch(nxt);                           // in flight while this image is metered
        wave_lds_fence();
        const uint32_t fi = first_frame + item * (uint32_t)kSuperFrames + lane;
        const bool live = fi < n_frames;
        const uint32_t fic = live ? fi : n_frames - 1u;
        const bool alaw = codec[fic % C] == IGDSP_PT_PCMA;
        uint32_t l = live ? (len ? min((uint32_t)len[fic], n) : n) : 0u;
        const uint32_t lm = alaw ? 0x80808080u : 0u;
        const uint32_t *row = img + lane * D;
        uint64_t s = 0;
        uint32_t peak = 0, bsum = 0;
        const uint32_t steps = (l + 3u) >> 2;                    // dwords this lane meters (its own loop bound)
        uint32_t i_first = 0;
        if (!RAGGED) {
            // dense frames: every lane walks the same D dwords, in units of two dwords = 8 samples.  Two-deep software pipeline as
            // in process_half: the eight LUT reads of unit u + 1 and the two image reads of unit u + 2 are in flight while unit
            // u is folded (reads past the frame's end are clamped to its last dword and never folded).
            const uint32_t units = D >> 1, last = D - 1u;
            uint2 ea[8], eb[8];
            uint32_t wa0, wa1, wb0, wb1;                          // image dwords of the unit in ea / eb
            auto issue = [&](uint32_t x0, uint32_t x1, uint2 (&e)[8]) {
                const uint32_t t0 = (x0 & 0x7F7F7F7Fu) | lm, t1 = (x1 & 0x7F7F7F7Fu) | lm;
                e[0] = lut_at(lut, t0, off, 0x0C0C0400u); e[1] = lut_at(lut, t0, off, 0x0C0C0500u); e[2] = lut_at(lut, t0, off, 0x0C0C0600u); e[3] = lut_at(lut, t0, off, 0x0C0C0700u);
                e[4] = lut_at(lut, t1, off, 0x0C0C0400u); e[5] = lut_at(lut, t1, off, 0x0C0C0500u); e[6] = lut_at(lut, t1, off, 0x0C0C0600u); e[7] = lut_at(lut, t1, off, 0x0C0C0700u);
            };
            uint32_t part = 0;
            auto fold = [&](uint32_t x0, uint32_t x1, const uint2 (&e)[8]) {
                bsum = __builtin_amdgcn_sad_u8(x0, 0u, bsum); bsum = __builtin_amdgcn_sad_u8(x1, 0u, bsum);
                part = part + e[0].x + e[1].x; part = part + e[2].x + e[3].x; part = part + e[4].x + e[5].x; part = part + e[6].x + e[7].x;
                peak = max(max(peak, e[0].y), e[1].y); peak = max(max(peak, e[2].y), e[3].y);
                peak = max(max(peak, e[4].y), e[5].y); peak = max(max(peak, e[6].y), e[7].y);
            };
            // image dwords travel two units ahead of the LUT reads that use them and are ISSUED before those: LDS returns in
            // order, so a wait for a dword pair must never sit behind the eight LUT reads issued after it
            const uint32_t lastu = units ? units - 1u : 0u;
            auto rd = [&](uint32_t uu, uint32_t &x0, uint32_t &x1) { const uint32_t *q = row + 2u * min(uu, lastu); x0 = q[0]; x1 = q[min(1u, last)]; };
            uint32_t wc0, wc1, wd0, wd1;
            rd(0u, wa0, wa1); rd(1u, wb0, wb1); rd(2u, wc0, wc1);
            issue(wa0, wa1, ea);
            uint32_t u = 0;
            for (; u + 2u <= units; u += 2u) {                     // unit u sits in ea (dwords wa), unit u + 1 is issued into eb (dwords wb)
                rd(u + 3u, wd0, wd1);
                issue(wb0, wb1, eb);
                __builtin_amdgcn_sched_barrier(0);
                fold(wa0, wa1, ea);
                rd(u + 4u, wa0, wa1);
                issue(wc0, wc1, ea);                              // unit u + 2
                __builtin_amdgcn_sched_barrier(0);
                fold(wb0, wb1, eb);
                { const uint32_t t0 = wa0, t1 = wa1; wa0 = wc0; wa1 = wc1; wb0 = wd0; wb1 = wd1; wc0 = t0; wc1 = t1; }
                if ((u & 2u) != 0u) { s += part; part = 0; }      // every 32 samples: 32 x 2^26 still fits 32 bits
            }
            if (u < units) { fold(wa0, wa1, ea); u += 1u; }       // an odd unit count leaves one issued unit in ea
            s += part;
            i_first = u << 1;                                      // the D % 2 dword left over takes the general loop below
        }
        for (uint32_t i0 = i_first; i0 < steps; i0 += 4u) {      // 16 samples per pass: (|x|/4)^2 < 2^26 each, the pass sum fits 32 bits
            uint32_t part = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t i = i0 + k;
                if (i < steps) {
                    uint32_t w = row[i];
                    const uint32_t nv = min(l - 4u * i, 4u);       // bytes of this dword inside the frame's length
                    const uint32_t t = (w & 0x7F7F7F7Fu) | lm;
                    const uint2 e0 = lut_at(lut, t, off, 0x0C0C0400u), e1 = lut_at(lut, t, off, 0x0C0C0500u);
                    const uint2 e2 = lut_at(lut, t, off, 0x0C0C0600u), e3 = lut_at(lut, t, off, 0x0C0C0700u);
                    if (nv == 4u) {
                        part += e0.x + e1.x + e2.x + e3.x;
                        peak = max(max(peak, e0.y), max(e1.y, max(e2.y, e3.y)));
                    } else {                                       // last dword of a ragged frame
                        part += e0.x + (nv > 1u ? e1.x : 0u) + (nv > 2u ? e2.x : 0u);
                        peak = max(max(peak, e0.y), max(nv > 1u ? e1.y : 0u, nv > 2u ? e2.y : 0u));
                        w &= (1u << (8u * nv)) - 1u;
                    }
                    bsum = __builtin_amdgcn_sad_u8(w, 0u, bsum);
                }
            }
            s += part;
        }
        // the reference's silence probe: payload bytes 28 / 38 / 48 of the lane's own frame
        bool probe = false;
        if (l > 48u) probe = ((row[7] & 0xFFu) == 0xD5u) && (((row[9] >> 16) & 0xFFu) == 0xD5u) && ((row[12] & 0xFFu) == 0xD5u);
        uint32_t bm = 0, fl = 0;
        uint4 rec = make_uint4(0u, 0u, 0u, (uint32_t)IGDSP_FLAG_EMPTY << 24);
        if (l != 0u) rec = pack_stats(s << 4, peak, bsum, l, alaw, probe, bm, fl);
        if (live) st_stream(reinterpret_cast<uint4 *>(stats + fi), rec);
        if (AGG) {
            const bool met = l != 0u;
            if (met) { a_sumsq += s << 4; a_samp += l; a_bm += bm; a_peak = max(a_peak, peak); }
            u_frames += (uint32_t)__builtin_popcountll(__ballot(met));
            u_sil += (uint32_t)__builtin_popcountll(__ballot(met && (fl & IGDSP_FLAG_SILENT) != 0u));
            u_clip += (uint32_t)__builtin_popcountll(__ballot(met && (fl & IGDSP_FLAG_CLIPPED) != 0u));
        }
        wave_lds_fence();
    }
    if (AGG && agg != nullptr) {
        __shared__ uint2 agg_slots[kWavesPerBlock * 4];
        const bool l0 = lane == 0u;
        agg_commit_block(agg, rank, agg_slots, n_waves, a_sumsq, (uint64_t)a_samp, l0 ? u_frames : 0u, l0 ? u_sil : 0u, l0 ? u_clip : 0u, a_bm, a_peak);
    }
}

// ============================================================================
// The reference's other frame sizes at full speed — k_meter_strided<Q, TAIL>: k_meter_chunk64's pipeline for dense frames of
// n = 16 Q + 4 T bytes (T = 0, 1, 2 tail dwords): Q = 10 -> 164 (the size the hook anticipates, roip_ed137.cpp:6561), 168;
// Q = 5 -> 80 (10 ms); Q = 15 -> 240 (30 ms); Q = 1 -> 16, 20, 24 (24: the other anticipated size).  A frame no longer starts
// on a 16-byte boundary, so piece q of frame f is fetched from f * n + 16 q with a dword-aligned 16-byte load (as the packed
// packet kernel does) and never straddles a frame.  With a tail (TAIL) every frame has one more piece, the frame's LAST 16
// bytes [n - 16, n): it overlaps piece Q - 1 (same cache lines, same load instruction: no extra memory traffic), rides
// through the expansion pipeline like the header pieces of k_meter_rtp64 (1 / (Q + 1) wasted LUT work) and hands its last two
// dwords RAW to the frame's lane through the strip; that lane expands the T tail dwords itself at fold time.  (A first
// version let the frame lane load its tail dwords from global memory: 64 scattered 4-byte requests per item re-fetched the
// lines — 0.48 of peak at n = 164 against 0.80 at n = 240.)  Item = 64 frames = Q + TAIL wave-wide loads, every piece
// register re-loaded from the next item the moment it is folded; block / device work queue as in k_meter_chunk64.
// ============================================================================
// STORE: the decoded int16 PCM goes out as well (pcm[F][C][n], dword aligned): every payload piece stores its 32 bytes as two
// dword-aligned 16-byte stores, the tail piece the 8 T bytes of the frame's tail samples; 12 waves (eight more live registers).
#ifndef IGDSP_SSTORE_WAVES
#define IGDSP_SSTORE_WAVES 0
#endif
// (the PCM-store variant, a 1 : 2 read : write mix, is fastest with few waves, as k_meter_chunk64<STORE>: 164-byte frames 12 / 8 / 6 / 5 / 4
// waves 0.7264 / 0.7133 / 0.7036 / 0.741 / 0.861 ms; 240: 10 / 8 / 6 / 5 / 4 0.9926 / 0.9865 / 0.980 / 0.9766 / 0.9975; 80: 12 / 10 / 8 / 6 0.3472 / 0.3462 / 0.3424 / 0.3402)
template <int QP, bool STORE = false> struct StridedGeom { static constexpr int kWaves = STORE ? (IGDSP_SSTORE_WAVES ? IGDSP_SSTORE_WAVES : (QP <= 2 ? 16 : 6)) : (QP <= 11 ? 16 : 12); };

template <int Q, bool TAIL, bool AGG, bool STORE = false>
__global__ __launch_bounds__((StridedGeom<Q + (TAIL ? 1 : 0), STORE>::kWaves * 64)) void k_meter_strided(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t n_frames, uint32_t n,
    igdsp_frame_stats *__restrict__ stats, igdsp_aggregate *agg, uint32_t rank, uint32_t *gqueue, int16_t *__restrict__ pcm = nullptr)
{
    static_assert(Q == 1 || Q >= 4, "the probe bytes 28 / 38 / 48 are taken from pieces 1 / 2 / 3");
    constexpr int QP = Q + (TAIL ? 1 : 0);                       // pieces per frame
    constexpr int kWaves = StridedGeom<QP, STORE>::kWaves;
    constexpr int kRow = QP | 1;                                 // strip row of a frame: an ODD number of 8-byte entries, so the fold's ds_read_b64 meet no bank twice
    constexpr int kStrip = kSuperFrames * kRow;                  // (rows of 8 entries = 16 banks put lanes l, l + 2, ... on the same banks: 8-way conflicts at n = 128)
    constexpr int kLutU2 = STORE ? kLutEntries : kLut32Words / 2;   // records only: the 4-byte m * m table (64 KiB as well: 256 raw-byte rows, igdsp_device.h); with PCM the 8-byte one
    __shared__ uint2 lds[kLutU2 + kWaves * kStrip + (STORE ? kWaves * 256 : 0)];        // LUT, strips (+ 2 KiB PCM transposition scratch per wave)
    __shared__ BlockQueue<kWaves> bq;
    __shared__ AggBlock aggb;
    uint32_t gb1 = 0;
    if (threadIdx.x == 0 && gqueue != nullptr) gb1 = atomicAdd(gqueue, 1u);
    if (STORE) fill_lut(lds); else fill_lut32(reinterpret_cast<uint32_t *>(lds));
    if (threadIdx.x == 0) { bq_init(bq, gqueue, gridDim.x, gb1); agg_block_init(aggb); }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint2 *strip = lds + kLutU2 + wave * kStrip;
    uint8_t *xpose = STORE ? reinterpret_cast<uint8_t *>(lds + kLutU2 + kWaves * kStrip + wave * 256) : nullptr;
    const uint32_t off = (lane & 31u) * (STORE ? 8u : 4u);
    const uint32_t *l32 = reinterpret_cast<const uint32_t *>(lds);
    const uint32_t T = (n - 16u * Q) >> 2;                       // tail dwords per frame (TAIL: 1 or 2), wave-uniform
    // per-lane piece constants, two pieces per register: frame of the item (6 bits) | probe shift << 8 (24 = none) | tail
    // piece << 13, in each 16-bit half.  The byte offset of piece j inside the item follows from the frame and the piece
    // number: f * n + (tail ? n - 16 : 16 q).
    constexpr int kPk = (QP + 1) / 2;
    constexpr bool kPoRegs = !STORE || QP <= 11;                 // the piece offsets stay in registers where they fit (with PCM at 15 pieces they do not)
    uint32_t pk[kPk];
    uint32_t po[kPoRegs ? QP : 1];
#pragma unroll
    for (int j = 0; j < kPk; ++j) pk[j] = 0;
#pragma unroll
    for (int j = 0; j < QP; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane, f = p / (uint32_t)QP, q = p - f * (uint32_t)QP;
        const uint32_t sh = q == 1u ? 0u : (q == 3u ? 8u : (q == 2u ? 16u : 24u));
        const bool tp = TAIL && q == (uint32_t)Q;
        pk[j >> 1] |= ((f << 2) | ((tp ? 24u : sh) << 8) | (tp ? 0x2000u : 0u)) << (16 * (j & 1));     // f * 4: also the ds_bpermute address of the frame's lane
        if (kPoRegs) po[j] = f * n + (tp ? n - 16u : 16u * q);
    }
    auto fr4_of = [&](int j) { return __builtin_amdgcn_ubfe(pk[j >> 1], 16 * (j & 1), 8); };
    auto fr_of = [&](int j) { return __builtin_amdgcn_ubfe(pk[j >> 1], 16 * (j & 1) + 2, 6); };
    auto ps_of = [&](int j) { return __builtin_amdgcn_ubfe(pk[j >> 1], 16 * (j & 1) + 8, 5); };
    auto tail_of = [&](int j) { return TAIL && ((pk[j >> 1] >> (16 * (j & 1) + 13)) & 1u) != 0u; };
    auto po_of = [&](int j) {                                    // byte offset of this lane's piece j inside an item
        if constexpr (kPoRegs) return po[j];
        else {
            const uint32_t f = fr_of(j), q = (uint32_t)j * 64u + lane - f * (uint32_t)QP;
            return f * n + (tail_of(j) ? n - 16u : 16u * q);
        }
    };
    const uint32_t G = gridDim.x;
    const uint32_t n_super = n_frames / kSuperFrames;           // the launcher hands over whole items only
    const uint64_t item_bytes = (uint64_t)kSuperFrames * n;
    uint64_t a_sumsq = 0;
    uint32_t a_bm = 0, a_peak = 0;
    uint32_t u_frames = 0, u_sil = 0, u_clip = 0;
    auto fetch_pt = [&](uint32_t sidx) { return (uint32_t)codec[(sidx * (uint32_t)kSuperFrames + lane) % C]; };
    // A queue slot stands for kPerSlot CONSECUTIVE items when items are small: the device-wide queue costs one device atomic
    // (~2 us of latency, prefetched one batch ahead) per batch of kWaves slots, and with 1.5-5 KiB items that atomic rate, not
    // the memory system, bounded the launch (n = 24: 0.39 of peak with one item per slot, 0.63 with a static schedule).
    constexpr uint32_t kPerSlot = STORE ? 1u : (QP <= 2 ? 8u : (QP <= 7 ? 2u : 1u));   // (with PCM the stores bound the launch: no gain measured)
    const uint32_t n_slots = (n_super + kPerSlot - 1u) / kPerSlot;
    // with a bulk output (PCM) the two halves of the batch range are visited alternately, as in k_meter_chunk64: igdsp_io_alloc
    // spreads a bulk buffer's halves over two memory classes and the writes should reach both at any moment (a batch = kWaves slots)
    const uint32_t n_batches = STORE ? (n_slots + (uint32_t)kWaves - 1u) / (uint32_t)kWaves : 0u;
    uint32_t g_slot = spread_batch(blockIdx.x, n_batches) * (uint32_t)kWaves + wave, g_k = 0;     // the wave's first slot is static
    auto grab = [&]() -> uint32_t {                              // next item id of this wave (>= n_super: none left)
        if (kPerSlot > 1u && g_k + 1u < kPerSlot && g_slot != 0xFFFFFFFFu) { ++g_k; return g_slot * kPerSlot + g_k; }
        g_slot = bq_grab(bq, gqueue, G, lane, n_batches);
        g_k = 0;
        if (g_slot >= n_slots) { g_slot = 0xFFFFFFFFu; return 0xFFFFFFFFu; }
        return g_slot * kPerSlot;
    };

    // Lookahead: at 16 - 24 bytes per frame an item is 1 - 1.5 KiB and ONE item of loads in flight per wave does not cover the
    // memory latency (16 waves x 1.5 KiB = 24 KiB per CU at n = 24, against ~150 KiB for n >= 160).  The record-only kernels for
    // those sizes keep TWO items in registers: while item i is expanded out of A (its registers refilled from item i + 2), item
    // i + 1 is already in flight into B (n = 24: 0.58 -> 0.60 of peak).  From four pieces per frame on it does not pay any more
    // (n = 64 / 80 / 96: 0.79 / 0.78 / 0.78 with one item ahead, 0.77 / 0.77 / 0.76 with two).
    constexpr bool kTwoAhead = !STORE && QP <= 2;
    const uint32_t sidx0 = g_slot * kPerSlot;                    // item 0 of the wave's first slot
    if (sidx0 < n_super) {
        // One item: expand it out of d (item sidx, codec id cur_pt of this lane's frame), re-load every piece register from item s_load
        // the moment it has been folded (s_load = 0 past the end: item 0 is L2-hot and the loads stay unconditional), draw one more
        // item id when asked (its LDS round trip hides under the fold); returns the codec id of this lane's frame in item s_load.
        auto step = [&](uint4 (&d)[QP], const uint32_t sidx, const uint32_t cur_pt, const uint32_t s_load, const bool do_grab,
                        uint32_t &s_after) __attribute__((always_inline)) -> uint32_t {
            const uint32_t f0 = sidx * kSuperFrames;
            const bool my_alaw = cur_pt == IGDSP_PT_PCMA;          // lane = frame; the pieces fetch it from here with ds_bpermute
            const uint8_t *nbase = payload + (uint64_t)s_load * item_bytes;
            const uint32_t nxt_pt = fetch_pt(s_load);
#pragma unroll
            for (int j = 0; j < kPk; ++j) asm volatile("" : "+v"(pk[j]));   // unpack per use: hoisted, the constants would take 3 QP registers
            {
                uint2 e[2][8];
                uint32_t wa[2], wb[2];
                uint32_t oj = off, lmj = 0;
                const uint32_t law_off = my_alaw ? 0x80u : 0u, law_mask = my_alaw ? 0x80808080u : 0u;
                auto issue = [&](int u) {
                    const int j = u >> 1, k = u & 1;
                    wa[k] = (u & 1) ? d[j].z : d[j].x;
                    wb[k] = (u & 1) ? d[j].w : d[j].y;
                    if (STORE) {
                        if (k == 0) lmj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)fr4_of(j), (int)law_mask);   // the law of this piece's frame, from the frame's own lane
                        const uint32_t ta = (wa[k] & 0x7F7F7F7Fu) | lmj, tb = (wb[k] & 0x7F7F7F7Fu) | lmj;
                        e[k][0] = lut_at(lds, ta, off, 0x0C0C0400u); e[k][1] = lut_at(lds, ta, off, 0x0C0C0500u);
                        e[k][2] = lut_at(lds, ta, off, 0x0C0C0600u); e[k][3] = lut_at(lds, ta, off, 0x0C0C0700u);
                        e[k][4] = lut_at(lds, tb, off, 0x0C0C0400u); e[k][5] = lut_at(lds, tb, off, 0x0C0C0500u);
                        e[k][6] = lut_at(lds, tb, off, 0x0C0C0600u); e[k][7] = lut_at(lds, tb, off, 0x0C0C0700u);
                    } else {                                     // e[][].x = m * m; `peak` tracks max(m * m) until the piece is complete
                        // the law of this piece's frame picks the row half: the frame's own lane holds it, one ds_bpermute fetches it
                        if (k == 0) oj = off | (uint32_t)__builtin_amdgcn_ds_bpermute((int)fr4_of(j), (int)law_off);
                        e[k][0].x = lut32_at(l32, wa[k], oj, 0x0C0C0400u); e[k][1].x = lut32_at(l32, wa[k], oj, 0x0C0C0500u);
                        e[k][2].x = lut32_at(l32, wa[k], oj, 0x0C0C0600u); e[k][3].x = lut32_at(l32, wa[k], oj, 0x0C0C0700u);
                        e[k][4].x = lut32_at(l32, wb[k], oj, 0x0C0C0400u); e[k][5].x = lut32_at(l32, wb[k], oj, 0x0C0C0500u);
                        e[k][6].x = lut32_at(l32, wb[k], oj, 0x0C0C0600u); e[k][7].x = lut32_at(l32, wb[k], oj, 0x0C0C0700u);
                    }
                };
                uint32_t sum = 0, peak = 0, bsum = 0;
                uint32_t o[8];
                issue(0);
#pragma unroll
                for (int u = 0; u < 2 * QP; ++u) {
                    const int j = u >> 1, k = u & 1;
                    if (u + 1 < 2 * QP) issue(u + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    bsum = __builtin_amdgcn_sad_u8(wa[k], 0u, bsum);
                    bsum = __builtin_amdgcn_sad_u8(wb[k], 0u, bsum);
                    sum = sum + e[k][0].x + e[k][1].x; sum = sum + e[k][2].x + e[k][3].x;
                    sum = sum + e[k][4].x + e[k][5].x; sum = sum + e[k][6].x + e[k][7].x;
                    if (STORE) {
                        peak = max(max(peak, e[k][0].y), e[k][1].y); peak = max(max(peak, e[k][2].y), e[k][3].y);
                        peak = max(max(peak, e[k][4].y), e[k][5].y); peak = max(max(peak, e[k][6].y), e[k][7].y);
                    } else {
                        peak = max(max(peak, e[k][0].x), e[k][1].x); peak = max(max(peak, e[k][2].x), e[k][3].x);
                        peak = max(max(peak, e[k][4].x), e[k][5].x); peak = max(max(peak, e[k][6].x), e[k][7].x);
                    }
                    if (STORE) {
                        o[4 * k + 0] = pack_pcm(wa[k], 0, e[k][0].y, e[k][1].y); o[4 * k + 1] = pack_pcm(wa[k], 2, e[k][2].y, e[k][3].y);
                        o[4 * k + 2] = pack_pcm(wb[k], 0, e[k][4].y, e[k][5].y); o[4 * k + 3] = pack_pcm(wb[k], 2, e[k][6].y, e[k][7].y);
                    }
                    if (k == 1) {                               // piece j complete
                        if (!STORE) peak = isqrt_m2(peak) << 2;             // max(m * m) -> |x| of the piece's loudest sample
                        uint2 ent = make_uint2(sum, peak | (bsum << 16) | probe_fail(d[j], 0xFFu << ps_of(j)));
                        if (tail_of(j)) ent = make_uint2(d[j].z, d[j].w);       // the frame's last two dwords, raw
                        strip[j * 64 + lane + (uint32_t)(kRow - QP) * fr_of(j)] = ent;       // piece p -> row p / QP, column p % QP
                        if (STORE) {
                            // The wave's PCM for row j is ONE contiguous run: piece p = 64 j + lane puts its 32 bytes (the tail piece:
                            // the 8 T bytes of the frame's tail samples) at S(p) = 2 n f + 32 q of the item's PCM, and S grows by
                            // exactly the previous piece's length from lane to lane.  Lanes park their bytes at S - S(lane 0) in a
                            // per-wave 2 KiB LDS scratch and read the run back 16 bytes per lane, so each store instruction writes
                            // 1 KiB contiguous instead of 16-byte pieces at 32-byte stride (0.58 -> see DESIGN 3.2b at n = 164).
                            const bool tl = tail_of(j);
                            const uint32_t fj = fr_of(j), qj = (uint32_t)j * 64u + lane - fj * (uint32_t)QP;
                            const uint32_t S = fj * 2u * n + (tl ? 32u * (uint32_t)Q : 32u * qj);
                            const uint32_t R = (uint32_t)__builtin_amdgcn_readfirstlane((int)S);
                            const uint32_t L = TAIL ? (uint32_t)__builtin_amdgcn_readlane((int)(S + (tl ? 8u * T : 32u)), 63) - R : 2048u;
                            uint2 *xp = reinterpret_cast<uint2 *>(xpose + (S - R));
                            if (!tl) {
                                xp[0] = make_uint2(o[0], o[1]); xp[1] = make_uint2(o[2], o[3]);
                                xp[2] = make_uint2(o[4], o[5]); xp[3] = make_uint2(o[6], o[7]);
                            } else if (T == 2u) {
                                xp[0] = make_uint2(o[4], o[5]); xp[1] = make_uint2(o[6], o[7]);
                            } else {
                                xp[0] = make_uint2(o[6], o[7]);
                            }
                            wave_lds_fence();
                            const uint4 r0 = reinterpret_cast<const uint4 *>(xpose)[lane], r1 = reinterpret_cast<const uint4 *>(xpose)[64u + lane];
                            wave_lds_fence();
                            uint8_t *ob = reinterpret_cast<uint8_t *>(pcm) + 2ull * ((uint64_t)sidx * item_bytes) + R;
                            u32x4_a4_t v0, v1;
                            v0.x = r0.x; v0.y = r0.y; v0.z = r0.z; v0.w = r0.w; v1.x = r1.x; v1.y = r1.y; v1.z = r1.z; v1.w = r1.w;
                            const uint32_t c0 = 16u * lane, c1 = 1024u + 16u * lane;
                            if (!TAIL || c0 + 16u <= L) *reinterpret_cast<u32x4_a4_t *>(ob + c0) = v0;
                            else if (c0 < L) { reinterpret_cast<uint32_t *>(ob + c0)[0] = r0.x; reinterpret_cast<uint32_t *>(ob + c0)[1] = r0.y; }
                            if (!TAIL || c1 + 16u <= L) *reinterpret_cast<u32x4_a4_t *>(ob + c1) = v1;
                            else if (c1 < L) { reinterpret_cast<uint32_t *>(ob + c1)[0] = r1.x; reinterpret_cast<uint32_t *>(ob + c1)[1] = r1.y; }
                        }
                        d[j] = ld16_dw(nbase + po_of(j));
                        sum = 0; peak = 0; bsum = 0;
                    }
                }
            }
            s_after = do_grab ? grab() : 0xFFFFFFFFu;
            wave_lds_fence();
            {
                const uint2 *row = strip + lane * kRow;            // the pieces of this lane's frame
                uint64_t s = 0;
                uint32_t peak = 0, bsum = 0, fail = 0, part = 0;
#pragma unroll
                for (int i = 0; i < Q; ++i) {
                    const uint2 v = row[i];
                    part += v.x;                                  // 30-bit piece sums: four fit 32 bits
                    if ((i & 3) == 3 || i == Q - 1) { s += part; part = 0; }
                    peak = max(peak, v.y & 0x7FFFu);
                    bsum += (v.y >> 16) & 0x7FFFu;
                    fail |= v.y;
                }
                if (TAIL) {                                        // the frame's tail dwords (bytes 16 Q .. n - 1), expanded by the frame's own lane
                    const uint2 tv = row[Q];
                    const uint32_t lm = my_alaw ? 0x80808080u : 0u;
                    const uint32_t tws[2] = {T == 2u ? tv.x : tv.y, tv.y};
                    uint32_t tail_m2 = 0;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        if ((uint32_t)t < T) {
                            const uint32_t w = tws[t];
                            if (STORE) {
                                const uint32_t tt = (w & 0x7F7F7F7Fu) | lm;
                                const uint2 e0 = lut_at(lds, tt, off, 0x0C0C0400u), e1 = lut_at(lds, tt, off, 0x0C0C0500u);
                                const uint2 e2 = lut_at(lds, tt, off, 0x0C0C0600u), e3 = lut_at(lds, tt, off, 0x0C0C0700u);
                                s += (uint64_t)(e0.x + e1.x + e2.x + e3.x);
                                peak = max(max(peak, e0.y), max(e1.y, max(e2.y, e3.y)));
                            } else {
                                const uint32_t ot = off | (lm & 0x80u);
                                const uint32_t q0 = lut32_at(l32, w, ot, 0x0C0C0400u), q1 = lut32_at(l32, w, ot, 0x0C0C0500u);
                                const uint32_t q2 = lut32_at(l32, w, ot, 0x0C0C0600u), q3 = lut32_at(l32, w, ot, 0x0C0C0700u);
                                s += (uint64_t)(q0 + q1 + q2 + q3);
                                tail_m2 = max(tail_m2, max(max(q0, q1), max(q2, q3)));
                            }
                            bsum = __builtin_amdgcn_sad_u8(w, 0u, bsum);
                        }
                    if (!STORE) peak = max(peak, isqrt_m2(tail_m2) << 2);
                }
                const bool probe = (Q >= 4) && (fail >> 31) == 0u;  // bytes 28 / 38 / 48 exist only when n > 48
                uint32_t bm, fl;
                st_stream(reinterpret_cast<uint4 *>(stats + (f0 + lane)), pack_stats(s << 4, peak, bsum, n, my_alaw, probe, bm, fl));
                if (AGG) {
                    a_sumsq += s << 4; a_bm += bm; a_peak = max(a_peak, peak);
                    u_frames += (uint32_t)kSuperFrames;
                    u_sil += (uint32_t)__builtin_popcountll(__ballot((fl & IGDSP_FLAG_SILENT) != 0u));
                    u_clip += (uint32_t)__builtin_popcountll(__ballot((fl & IGDSP_FLAG_CLIPPED) != 0u));
                }
            }
            wave_lds_fence();
            return nxt_pt;
        };
        auto load_item = [&](uint4 (&d)[QP], const uint32_t s) __attribute__((always_inline)) {
            const uint8_t *b0 = payload + (uint64_t)s * item_bytes;
#pragma unroll
            for (int j = 0; j < QP; ++j) d[j] = ld16_dw(b0 + po_of(j));
        };
        if (!kTwoAhead) {
            uint4 d[QP];
            uint32_t sidx = sidx0, cur_pt = fetch_pt(sidx);
            load_item(d, sidx);
            uint32_t s_next = grab();
            for (;;) {
                const bool has_next = s_next < n_super;
                uint32_t s_after;
                const uint32_t nxt_pt = step(d, sidx, cur_pt, has_next ? s_next : 0u, has_next, s_after);
                if (!has_next) break;
                sidx = s_next;
                s_next = s_after;
                cur_pt = nxt_pt;
            }
        } else {
            uint4 A[QP], B[QP];
            uint32_t x0 = sidx0, p0 = fetch_pt(x0);
            load_item(A, x0);
            uint32_t x1 = grab();
            bool v1 = x1 < n_super;
            uint32_t p1 = fetch_pt(v1 ? x1 : 0u);
            load_item(B, v1 ? x1 : 0u);
            uint32_t x2 = v1 ? grab() : 0xFFFFFFFFu;
            for (;;) {
                const bool v2 = x2 < n_super;
                uint32_t x3, x4;
                const uint32_t p2 = step(A, x0, p0, v2 ? x2 : 0u, v2, x3);      // item x0 out of A; A refilled from x2; x3 drawn
                if (!v1) break;
                const bool v3 = x3 < n_super;
                const uint32_t p3 = step(B, x1, p1, v3 ? x3 : 0u, v3, x4);      // item x1 out of B; B refilled from x3; x4 drawn
                if (!v2) break;
                x0 = x2; p0 = p2; x1 = x3; p1 = p3; v1 = v3; x2 = x4;
            }
        }
    }
    wave_exit(AGG ? agg : nullptr, rank, aggb, (uint32_t)kWaves, lane, gqueue, G, a_sumsq, lane == 0u ? (uint64_t)u_frames * n : 0ull,
              a_bm, a_peak, u_frames, u_sil, u_clip);
}

// ============================================================================
// Tiny frames — k_meter_tiny<N4>: dense frames of n = 4 N4 bytes, N4 = 4 .. 8 (16 .. 32 bytes: the 24-byte payload the hook
// anticipates, roip_ed137.cpp:6562, and its neighbours), records only.  At these sizes a frame is smaller than two 16-byte
// pieces, so the piece / strip machinery of k_meter_strided spends most of its instructions on bookkeeping (two wave-wide loads,
// a strip round trip and a tail hand-over per 1.5 KiB item: 0.60 of peak at n = 24).  Here a LANE owns a FRAME outright: it loads
// its own n bytes (one dword-aligned 16-byte load + the rest), expands them through the 4-byte m * m table (fill_lut32: one
// ds_read_b32 per sample, the law is one bit of the lane's table offset), keeps sum / max / byte-sum in three registers — 32
// samples x 2^26 still fit 32 bits — and stores its 16-byte record: 64 lanes = one 1 KiB store, no LDS hand-off at all.  No frame
// is long enough for the silence probe (bytes 28 / 38 / 48 need n > 48).  Items of 64 frames; a queue slot stands for kTinySlot
// consecutive items (four: the slots are what the block / device queue balances); kTinyDepth items of loads stay in flight per wave across slot boundaries.
// ============================================================================
constexpr int kTinyWaves = 16, kTinyDepth = 4;
// items per queue slot (>= kTinyDepth: the prologue).  16 made a wave's share two slots, i.e. static in effect; 32 / 16 / 8 / 4 items: 0.0683 /
// 0.0693 / 0.0682 / 0.0676 ms at 24-byte frames, 65 536 x 128 (same-box A/B builds, late round 3)
#ifndef IGDSP_TINY_SLOT
#define IGDSP_TINY_SLOT 4
#endif
constexpr uint32_t kTinySlot = IGDSP_TINY_SLOT;

template <int N4, bool AGG>
__global__ __launch_bounds__(kTinyWaves * 64) void k_meter_tiny(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t n_frames,
    igdsp_frame_stats *__restrict__ stats, igdsp_aggregate *agg, uint32_t rank, uint32_t *gqueue)
{
    static_assert(N4 >= 4 && N4 <= 8, "16 .. 32-byte frames");
    constexpr uint32_t n = 4u * N4;
    __shared__ uint32_t l32[kLut32Words];
    __shared__ BlockQueue<kTinyWaves> bq;
    __shared__ AggBlock aggb;
    uint32_t gb1 = 0;
    if (threadIdx.x == 0 && gqueue != nullptr) gb1 = atomicAdd(gqueue, 1u);

    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t G = gridDim.x;
    const uint32_t n_items = n_frames / kSuperFrames;            // the launcher hands over whole items only
    const uint32_t n_slots = (n_items + kTinySlot - 1u) / kTinySlot;
    const uint32_t off = (lane & 31u) * 4u;
    uint64_t a_sumsq = 0;
    uint32_t a_bm = 0, a_peak = 0, u_frames = 0, u_sil = 0, u_clip = 0;

    // the wave's item stream: slot ids from the block / device queue, kTinySlot consecutive items per slot (0xFFFFFFFF: none left)
    // At these sizes the records are 40 % of the traffic — the launch's BULK output — so the slot batches are visited alternately from
    // the two halves of the launch (spread_batch): a record buffer that igdsp_io_alloc spread over two memory classes (IGDSP_IO_BULK) is
    // then written in both at any moment (a 1 : 1 mix into one class streams at 0.66-0.68 of peak, into two at 0.78: DESIGN.md 7).
    const uint32_t n_batches = (n_slots + (uint32_t)kTinyWaves - 1u) / (uint32_t)kTinyWaves;
    uint32_t g_slot = spread_batch(blockIdx.x, n_batches) * (uint32_t)kTinyWaves + wave, g_k = 0;      // first slot static (batch = blockIdx, slot = wave)
    if (g_slot >= n_slots) g_slot = 0xFFFFFFFFu;
    auto next_item = [&]() -> uint32_t {
        if (g_slot == 0xFFFFFFFFu) return 0xFFFFFFFFu;
        if (g_k == kTinySlot) {
            g_slot = bq_grab(bq, gqueue, G, lane, n_batches);
            g_k = 0;
            if (g_slot >= n_slots) { g_slot = 0xFFFFFFFFu; return 0xFFFFFFFFu; }
        }
        const uint32_t it = g_slot * kTinySlot + g_k++;
        if (it >= n_items) { g_slot = 0xFFFFFFFFu; return 0xFFFFFFFFu; }       // the last slot may be short
        return it;
    };
    struct Item { uint4 a; uint32_t b[N4 > 4 ? N4 - 4 : 1]; uint32_t pt; };
    auto fetch = [&](uint32_t it, Item &x) {                     // it == none: item 0 again (L2-hot), so that no load is conditional
        const uint32_t fi = (it == 0xFFFFFFFFu ? 0u : it) * (uint32_t)kSuperFrames + lane;
        const uint8_t *p = payload + (uint64_t)fi * n;
        x.a = ld16_dw(p);
#pragma unroll
        for (int k = 4; k < N4; ++k) x.b[k - 4] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p) + k);
        x.pt = codec[fi % C];
    };
    // the first kTinyDepth items of a wave come from its static first slot (kTinySlot >= kTinyDepth: no queue needed yet), so their
    // loads go out BEFORE the table is filled: at 68 us per launch the first loads' latency is worth hiding under the fill
    static_assert(kTinySlot >= (uint32_t)kTinyDepth, "the prologue must not touch the block queue");
    Item q[kTinyDepth];
    uint32_t ids[kTinyDepth];
#ifndef IGDSP_TINY_LATE
#pragma unroll
    for (int d = 0; d < kTinyDepth; ++d) { ids[d] = next_item(); fetch(ids[d], q[d]); }
#endif
    fill_lut32(l32);
    if (threadIdx.x == 0) { bq_init(bq, gqueue, gridDim.x, gb1); agg_block_init(aggb); }
    __syncthreads();
#ifdef IGDSP_TINY_LATE            // A/B builds only: the first loads after the table fill, as before
#pragma unroll
    for (int d = 0; d < kTinyDepth; ++d) { ids[d] = next_item(); fetch(ids[d], q[d]); }
#endif
    auto step = [&](Item &slot, uint32_t &id) __attribute__((always_inline)) -> bool {
        const uint32_t it = id;
        if (it == 0xFFFFFFFFu) return false;                     // ids are handed out in order: the first missing one ends the stream
        const Item x = slot;
        id = next_item();
        fetch(id, slot);                                         // the register set is free again: kTinyDepth items ahead
        const bool alaw = x.pt == IGDSP_PT_PCMA;
        const uint32_t oj = off | (alaw ? 0x80u : 0u);
        uint32_t sum = 0, mx = 0, bsum = 0;
        auto dword = [&](uint32_t w) {
            const uint32_t q0 = lut32_at(l32, w, oj, 0x0C0C0400u), q1 = lut32_at(l32, w, oj, 0x0C0C0500u);
            const uint32_t q2 = lut32_at(l32, w, oj, 0x0C0C0600u), q3 = lut32_at(l32, w, oj, 0x0C0C0700u);
            sum = sum + q0 + q1; sum = sum + q2 + q3;
            mx = max(max(mx, q0), q1); mx = max(max(mx, q2), q3);
            bsum = __builtin_amdgcn_sad_u8(w, 0u, bsum);
        };
        dword(x.a.x); dword(x.a.y); dword(x.a.z); dword(x.a.w);
#pragma unroll
        for (int k = 4; k < N4; ++k) dword(x.b[k - 4]);
        const uint32_t peak = isqrt_m2(mx) << 2;
        uint32_t bm, fl;
        st_stream(reinterpret_cast<uint4 *>(stats + ((uint64_t)it * (uint32_t)kSuperFrames + lane)),
                  pack_stats((uint64_t)sum << 4, peak, bsum, n, alaw, false, bm, fl));
        if (AGG) {
            a_sumsq += (uint64_t)sum << 4; a_bm += bm; a_peak = max(a_peak, peak);
            u_frames += (uint32_t)kSuperFrames;
            u_sil += (uint32_t)__builtin_popcountll(__ballot((fl & IGDSP_FLAG_SILENT) != 0u));
            u_clip += (uint32_t)__builtin_popcountll(__ballot((fl & IGDSP_FLAG_CLIPPED) != 0u));
        }
        return true;
    };
    static_assert(kTinyDepth == 4, "the loop below names the four register sets");
    while (step(q[0], ids[0]) && step(q[1], ids[1]) && step(q[2], ids[2]) && step(q[3], ids[3])) {}
    wave_exit(AGG ? agg : nullptr, rank, aggb, (uint32_t)kTinyWaves, lane, gqueue, G, a_sumsq, lane == 0u ? (uint64_t)u_frames * n : 0ull,
              a_bm, a_peak, u_frames, u_sil, u_clip);
}

// Kernel attributes are per DEVICE: k_meter_image asks for more than 64 KiB of dynamic LDS (its frame images), which every device
// that runs it has to be told once.  igdsp_create calls this with its device current (round 2 kept one flag per process, so in a
// process with one context per GPU every device after the first would have had its image launches rejected).
hipError_t init_device_attributes()
{
    const int lim = 160 * 1024 - 2048 - kLutEntries * 8;
    const void *fns[] = {reinterpret_cast<const void *>(&k_meter_image<true, true>), reinterpret_cast<const void *>(&k_meter_image<false, true>),
                         reinterpret_cast<const void *>(&k_meter_image<true, false>), reinterpret_cast<const void *>(&k_meter_image<false, false>)};
    for (const void *f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_decode_meter(const LaunchCfg &cfg, int variant, const uint8_t *payload, const uint8_t *codec,
                               const uint16_t *len, uint32_t C, uint32_t F, uint32_t n, igdsp_frame_stats *stats,
                               int16_t *pcm, igdsp_aggregate *agg, uint32_t rank, hipStream_t s)
{
    uint32_t *gq = cfg.gqueue;
    const uint64_t n_frames64 = (uint64_t)C * F;
    if (n_frames64 == 0) return hipSuccess;
    const uint32_t n_frames = (uint32_t)n_frames64;
    const bool chunk_ok = (n == (uint32_t)kFrame) && (len == nullptr) &&
                          ((reinterpret_cast<uintptr_t>(payload) & 15u) == 0u) &&
                          (pcm == nullptr || (reinterpret_cast<uintptr_t>(pcm) & 15u) == 0u) &&
                          ((reinterpret_cast<uintptr_t>(stats) & 15u) == 0u);
    // tuned path takes the whole super-chunks (64 frames); the < 64 remaining frames, and every shape it
    // does not cover, go through the general wave-per-frame kernel on the same stream.
    uint32_t done = 0;
    if (variant == 3 && chunk_ok && pcm == nullptr && n_frames >= (uint32_t)kSuperFrames) {
        const uint32_t n_super = n_frames / kSuperFrames;
        done = n_super * kSuperFrames;
        const uint32_t grid = blocks_for(n_super, kFatWaves, (uint32_t)cfg.compute_units);
        if (agg) hipLaunchKernelGGL((k_meter_fat<true>), dim3(grid), dim3(kFatWaves * 64), 0, s, payload, codec, C, done, stats, agg, rank);
        else     hipLaunchKernelGGL((k_meter_fat<false>), dim3(grid), dim3(kFatWaves * 64), 0, s, payload, codec, C, done, stats, agg, rank);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    } else if (variant != 1 && chunk_ok && n_frames >= (uint32_t)kSuperFrames) {
        const uint32_t n_super = n_frames / kSuperFrames;
        done = n_super * kSuperFrames;
        uint64_t *nodiag = nullptr;
        if (pcm) {
            constexpr int w = ChunkGeom<true>::kWaves;
            const uint32_t grid = blocks_for(n_super, w, (uint32_t)cfg.compute_units);
            if (agg) hipLaunchKernelGGL((k_meter_chunk64<true, true>), dim3(grid), dim3(w * 64), 0, s, payload, codec, C, done, stats, pcm, agg, rank, nodiag, gq);
            else     hipLaunchKernelGGL((k_meter_chunk64<true, false>), dim3(grid), dim3(w * 64), 0, s, payload, codec, C, done, stats, pcm, agg, rank, nodiag, gq);
        } else {
            const uint32_t grid = blocks_for(n_super, kWavesPerBlock, (uint32_t)cfg.compute_units);
            if (agg) hipLaunchKernelGGL((k_meter_chunk64<false, true>), dim3(grid), dim3(kBlockThreads), 0, s, payload, codec, C, done, stats, pcm, agg, rank, nodiag, gq);
            else     hipLaunchKernelGGL((k_meter_chunk64<false, false>), dim3(grid), dim3(kBlockThreads), 0, s, payload, codec, C, done, stats, pcm, agg, rank, nodiag, gq);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // 16 .. 32-byte frames, records only: a lane per frame (k_meter_tiny)
    if (done == 0 && variant != 1 && len == nullptr && pcm == nullptr && (n & 3u) == 0u && n >= 16u && n <= 32u && n_frames >= (uint32_t)kSuperFrames &&
        ((reinterpret_cast<uintptr_t>(payload) & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(stats) & 15u) == 0u) && std::getenv("IGDSP_NO_TINY") == nullptr) {
        const uint32_t n_items = n_frames / kSuperFrames;
        const uint32_t whole = n_items * kSuperFrames;
        const uint32_t grid = blocks_for((n_items + kTinySlot - 1u) / kTinySlot, kTinyWaves, (uint32_t)cfg.compute_units);
#define IGDSP_TINY(NV)                                                                                                                              \
        if ((n >> 2) == NV) {                                                                                                                        \
            if (agg) hipLaunchKernelGGL((k_meter_tiny<NV, true>), dim3(grid), dim3(kTinyWaves * 64), 0, s, payload, codec, C, whole, stats, agg, rank, gq);   \
            else     hipLaunchKernelGGL((k_meter_tiny<NV, false>), dim3(grid), dim3(kTinyWaves * 64), 0, s, payload, codec, C, whole, stats, agg, rank, gq);  \
            done = whole;                                                                                                                            \
        }
        IGDSP_TINY(4) IGDSP_TINY(5) IGDSP_TINY(6) IGDSP_TINY(7) IGDSP_TINY(8)
#undef IGDSP_TINY
        if (done) { hipError_t e = hipGetLastError(); if (e != hipSuccess) return e; }
    }
    // dense frames of 16 Q + 4 T bytes, Q in {1, 4, 5, 6, 8, 10, 12, 15}, T <= 2 (the reference's 164 / 24 and the 5 ms multiples
    // up to 240) keep the chunk pipeline: k_meter_strided.  With PCM output: the reference's own sizes (24, 80, 164 / 168, 240).
    if (done == 0 && variant != 1 && len == nullptr && (n & 3u) == 0u && n_frames >= (uint32_t)kSuperFrames &&   // (160-byte frames land here only when their buffer is not 16-byte aligned)
        ((n >> 2) & 3u) != 3u && n >= 16u && ((reinterpret_cast<uintptr_t>(pcm) & 3u) == 0u) &&
        ((reinterpret_cast<uintptr_t>(payload) & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(stats) & 15u) == 0u) && std::getenv("IGDSP_NO_STRIDED") == nullptr) {
        const uint32_t Qn = n >> 4;
        const bool tail = (n & 15u) != 0u;
        const uint32_t n_super = n_frames / kSuperFrames;
        const uint32_t whole = n_super * kSuperFrames;
#define IGDSP_STRIDED(QV, TV)                                                                                                                         \
        if (Qn == QV && tail == TV && pcm == nullptr) {                                                                                               \
            constexpr int w = StridedGeom<QV + (TV ? 1 : 0)>::kWaves;                                                                                 \
            const uint32_t grid = blocks_for(n_super, w, (uint32_t)cfg.compute_units);                                                                \
            if (agg) hipLaunchKernelGGL((k_meter_strided<QV, TV, true>), dim3(grid), dim3(w * 64), 0, s, payload, codec, C, whole, n, stats, agg, rank, gq, pcm);   \
            else     hipLaunchKernelGGL((k_meter_strided<QV, TV, false>), dim3(grid), dim3(w * 64), 0, s, payload, codec, C, whole, n, stats, agg, rank, gq, pcm);  \
            done = whole;                                                                                                                             \
        }
#define IGDSP_STRIDED_PCM(QV, TV)                                                                                                                     \
        if (Qn == QV && tail == TV && pcm != nullptr) {                                                                                               \
            constexpr int w = StridedGeom<QV + (TV ? 1 : 0), true>::kWaves;                                                                           \
            const uint32_t grid = blocks_for(n_super, w, (uint32_t)cfg.compute_units);                                                                \
            if (agg) hipLaunchKernelGGL((k_meter_strided<QV, TV, true, true>), dim3(grid), dim3(w * 64), 0, s, payload, codec, C, whole, n, stats, agg, rank, gq, pcm);   \
            else     hipLaunchKernelGGL((k_meter_strided<QV, TV, false, true>), dim3(grid), dim3(w * 64), 0, s, payload, codec, C, whole, n, stats, agg, rank, gq, pcm);  \
            done = whole;                                                                                                                             \
        }
        IGDSP_STRIDED(1, false) IGDSP_STRIDED(1, true) IGDSP_STRIDED(4, false) IGDSP_STRIDED(4, true) IGDSP_STRIDED(5, false) IGDSP_STRIDED(5, true)
        IGDSP_STRIDED(6, false) IGDSP_STRIDED(6, true) IGDSP_STRIDED(8, false) IGDSP_STRIDED(8, true) IGDSP_STRIDED(10, false) IGDSP_STRIDED(10, true)
        IGDSP_STRIDED(12, false) IGDSP_STRIDED(12, true) IGDSP_STRIDED(15, false)   // (15, true) = 244 / 248 bytes: 16 pieces x 12 waves of strip do not fit
        IGDSP_STRIDED_PCM(1, true) IGDSP_STRIDED_PCM(5, false) IGDSP_STRIDED_PCM(10, true) IGDSP_STRIDED_PCM(10, false) IGDSP_STRIDED_PCM(15, false)
#undef IGDSP_STRIDED
#undef IGDSP_STRIDED_PCM
        if (done) { hipError_t e = hipGetLastError(); if (e != hipSuccess) return e; }
    }
    if (done < n_frames) {
        // what the tuned n == 160 kernel does not take: other frame sizes, ragged lengths, the < 64-frame tail.  Meter-only
        // work with n % 4 == 0 goes through the LDS-image kernel (every lane meters one frame); PCM output, n % 4 != 0 and
        // unaligned buffers through the literal wave-per-frame kernel.
        const bool image_ok = variant != 1 && pcm == nullptr && (n & 3u) == 0u && ((reinterpret_cast<uintptr_t>(stats) & 15u) == 0u) &&
                              ((reinterpret_cast<uintptr_t>(payload) & 3u) == 0u) && n_frames - done >= 16u;
        if (image_ok) {
            const uint32_t img = (uint32_t)kSuperFrames * n;
            const uint32_t lut_bytes = (uint32_t)kLutEntries * 8u;
            uint32_t waves = std::max(1u, std::min(kImgMaxWaves, (160u * 1024u - lut_bytes - 2048u) / img));
            if (const char *e = std::getenv("IGDSP_IMG_WAVES")) waves = std::max(1u, std::min(waves, (uint32_t)std::atoi(e)));   // experiments
            const uint32_t items = (n_frames - done + (uint32_t)kSuperFrames - 1u) / (uint32_t)kSuperFrames;
            const uint32_t grid = blocks_for(items, waves, (uint32_t)cfg.compute_units);
            const size_t smem = (size_t)waves * img;              // dynamic part: the images (the LUT is static)
            const dim3 g3(grid), b3(waves * 64u);
            if (len) {
                if (agg) hipLaunchKernelGGL((k_meter_image<true, true>), g3, b3, smem, s, payload, codec, len, C, done, n_frames, n, stats, agg, rank);
                else     hipLaunchKernelGGL((k_meter_image<false, true>), g3, b3, smem, s, payload, codec, len, C, done, n_frames, n, stats, agg, rank);
            } else {
                if (agg) hipLaunchKernelGGL((k_meter_image<true, false>), g3, b3, smem, s, payload, codec, len, C, done, n_frames, n, stats, agg, rank);
                else     hipLaunchKernelGGL((k_meter_image<false, false>), g3, b3, smem, s, payload, codec, len, C, done, n_frames, n, stats, agg, rank);
            }
        } else {
            const uint32_t grid = blocks_for((n_frames - done + 7) / 8, 4, (uint32_t)cfg.compute_units * 8u);
            hipLaunchKernelGGL(k_meter_wave_per_frame, dim3(grid), dim3(256), 0, s, payload, codec, len, C, done, n_frames, n, stats, pcm, agg, rank);
        }
    }
    return hipGetLastError();
}

hipError_t launch_diag_chunk32(const LaunchCfg &cfg, const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F,
                               igdsp_frame_stats *stats, uint64_t *diag, hipStream_t s)
{
    const uint32_t n_super = (C * F) / kSuperFrames;
    const uint32_t n_frames = n_super * kSuperFrames;
    const uint32_t grid = blocks_for(n_super, kWavesPerBlock, (uint32_t)cfg.compute_units);
    hipLaunchKernelGGL((k_meter_chunk64<false, false, true>), dim3(grid), dim3(kBlockThreads), 0, s, payload, codec, C, n_frames,
                       stats, (int16_t *)nullptr, (igdsp_aggregate *)nullptr, 0u, diag, cfg.gqueue);
    return hipGetLastError();
}


}  // namespace igdsp
