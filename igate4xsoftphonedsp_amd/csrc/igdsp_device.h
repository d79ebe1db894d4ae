// igdsp_device.h — device-side building blocks shared by the kernel translation units (igdsp_k_*.hip): G.711 expansion
// formulas, streaming load / store helpers (global and buffer form), the launch-aggregate commit, record packing, the
// 32-replica expansion LUT, probe helpers, the persistent-block work queue.  Header-only, namespace igdsp.
//
// Reference semantics the kernels reproduce (all citations /root/reference):
//   G.711 expansion / compression : performed by pjmedia around adapter->stream_rtp_cb (TransportAdapter.cpp:301) / before
//       transport_send_rtp (TransportAdapter.cpp:635); ITU-T G.711.
//   byte_mean "audioLevel"        : roip_ed137.cpp:6564-6568, 6513-6517.
//   silence probe                 : TransportAdapter.cpp:657-673.
//   hold / window aggregate       : Functions.cpp:2126-2145, 2155-2167.
#pragma once
#include "igdsp_internal.h"

#include <algorithm>
#include <cstdlib>

namespace igdsp {

static inline uint32_t blocks_for(uint64_t items, uint32_t per_block, uint32_t cap)
{
    uint64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (uint32_t)(b > cap ? cap : b);
}

// ----------------------------------------------------------------------------
// G.711 expansion magnitude by the ITU segment formula (used to build the LDS
// tables in-kernel; no table ever comes from host memory).
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ulaw_abs(uint32_t code)
{
    const uint32_t u = ~code & 0x7Fu;
    return ((((u & 15u) * 2u + 33u) << (u >> 4)) - 33u) << 2;
}

__device__ __forceinline__ uint32_t alaw_abs(uint32_t code)
{
    const uint32_t a = (code ^ 0x55u) & 0x7Fu;
    const uint32_t s = a >> 4, q = a & 15u;
    const uint32_t m = (s == 0u) ? (q * 2u + 1u) : ((q * 2u + 33u) << (s - 1u));
    return m << 3;
}

// streaming (read-once) 16-byte load: native vector type so the nontemporal builtin accepts it
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream(const uint4 *p)
{
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// 16 bytes at dword (not 16-byte) alignment: gfx950 global loads only need dword alignment for dwordx4 (measured < 1 %
// slower than aligned ones in a bare stream, tools/misaligned_loads.py).  Nontemporal like every other streaming load here.
typedef uint32_t u32x4_a4_t __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 ld16_dw(const uint8_t *p)
{
    const u32x4_a4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// write-once 16-byte store (records / PCM are never re-read by this launch)
#ifndef IGDSP_NT_STORE
#define IGDSP_NT_STORE 0      // cached stores measured ~1 % faster than nontemporal for the 1 KiB record blocks
#endif
__device__ __forceinline__ void st_stream(uint4 *p, const uint4 v)
{
#if IGDSP_NT_STORE
    u32x4_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t *>(p));
#else
    *p = v;
#endif
}

// Buffer-instruction form of the streaming accesses: a wave-uniform 128-bit descriptor (SGPRs) + one 32-bit lane offset
// (lane * 16) + a scalar offset.  The compiler never forms SGPR-base addressing for global_load / global_store on this path
// (every address is a per-lane 64-bit VGPR pair and 10 KiB spans need several of them), so the kernels that are short of
// registers describe their windows themselves.  Raw buffer, no swizzle, 32-bit data format; the range check is switched off
// by the largest record count (every offset used is < 2^32 because the base is re-seated per item / frame).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xFFFFFFFFu, 0x00020000);
}
// the same with a range: raw buffer accesses at an offset >= bytes are dropped by the hardware (stores) / return zero (loads),
// which lets a lane opt out of a store without a branch
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_ranged(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_ld_stream(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2);          // aux 2 = nt
    return make_uint4(v.x, v.y, v.z, v.w);
}
#ifndef IGDSP_RTS_LOAD_AUX
#define IGDSP_RTS_LOAD_AUX 2      // A/B: cache policy of k_roundtrip_strided's dword-aligned input pieces (2 = nt, 0 = default)
#endif
__device__ __forceinline__ uint4 buf_ld_pieces(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, IGDSP_RTS_LOAD_AUX);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, const uint4 v)
{
    u32x4_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)voff, (int)soff, 0);
}

__device__ __forceinline__ uint32_t full_scale(bool alaw) { return alaw ? 32256u : 32124u; }

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl_xor((int)lo, m, 64);
    hi = (uint32_t)__shfl_xor((int)hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Launch aggregate: wave butterfly -> one LDS slot per wave -> wave 0 folds the block -> ONE set of
// device-scope integer atomics per BLOCK (exact, order-independent u64 add / max).  Same-line atomics
// serialise at roughly 90 per microsecond, so per-wave commits (4096 waves x 7 words) cost ~0.3 ms;
// per-block commits keep it to a few microseconds that overlap with other blocks' tails.
// `slots` = nwaves x 4 uint2 of LDS.  Must be reached by every thread of the block.
__device__ __forceinline__ void agg_commit_block(igdsp_aggregate *agg, uint32_t rank, uint2 *slots, uint32_t nwaves,
                                                 uint64_t sumsq, uint64_t samples, uint32_t frames, uint32_t n_silent,
                                                 uint32_t n_clipped, uint32_t bm_sum, uint32_t peak)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sumsq += shfl_xor_u64(sumsq, m);
        samples += shfl_xor_u64(samples, m);
        frames += (uint32_t)__shfl_xor((int)frames, m, 64);
        n_silent += (uint32_t)__shfl_xor((int)n_silent, m, 64);
        n_clipped += (uint32_t)__shfl_xor((int)n_clipped, m, 64);
        bm_sum += (uint32_t)__shfl_xor((int)bm_sum, m, 64);
        peak = max(peak, (uint32_t)__shfl_xor((int)peak, m, 64));
    }
    __syncthreads();                       // every wave is done with its LDS strip
    if (lane == 0) {
        slots[wave * 4 + 0] = make_uint2((uint32_t)sumsq, (uint32_t)(sumsq >> 32));
        slots[wave * 4 + 1] = make_uint2((uint32_t)samples, (uint32_t)(samples >> 32));
        slots[wave * 4 + 2] = make_uint2(frames, n_silent);
        slots[wave * 4 + 3] = make_uint2(n_clipped | 0u, bm_sum);
    }
    // peak rides in a fifth word: reuse slot 2/3 would overflow nothing, keep it simple and separate
    __shared__ uint32_t peak_slots[kWavesPerBlock];
    if (lane == 0) peak_slots[wave] = peak;
    __syncthreads();
    if (wave == 0) {
        uint64_t s = 0, sm = 0;
        uint32_t fr = 0, sil = 0, cl = 0, bm = 0, pk = 0;
        if (lane < nwaves) {
            const uint2 a = slots[lane * 4 + 0], b = slots[lane * 4 + 1], c = slots[lane * 4 + 2], d = slots[lane * 4 + 3];
            s = ((uint64_t)a.y << 32) | a.x; sm = ((uint64_t)b.y << 32) | b.x;
            fr = c.x; sil = c.y; cl = d.x; bm = d.y; pk = peak_slots[lane];
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {   // nwaves <= 16
            s += shfl_xor_u64(s, m); sm += shfl_xor_u64(sm, m);
            fr += (uint32_t)__shfl_xor((int)fr, m, 64); sil += (uint32_t)__shfl_xor((int)sil, m, 64);
            cl += (uint32_t)__shfl_xor((int)cl, m, 64); bm += (uint32_t)__shfl_xor((int)bm, m, 64);
            pk = max(pk, (uint32_t)__shfl_xor((int)pk, m, 64));
        }
        if (lane == 0 && fr != 0) {
            atomicAdd((unsigned long long *)&agg->sumsq, (unsigned long long)s);
            atomicAdd((unsigned long long *)&agg->samples, (unsigned long long)sm);
            atomicAdd((unsigned long long *)&agg->frames, (unsigned long long)fr);
            atomicAdd((unsigned long long *)&agg->n_silent, (unsigned long long)sil);
            atomicAdd((unsigned long long *)&agg->n_clipped, (unsigned long long)cl);
            atomicAdd((unsigned long long *)&agg->byte_mean_sum, (unsigned long long)bm);
            atomicMax((unsigned long long *)&agg->peak_slot[rank & (IGDSP_AGG_MAX_RANKS - 1)], (unsigned long long)pk);
        }
    }
}

// the 16-byte record as one dwordx4 store: {sumsq lo, sumsq hi, rms bits, peak | byte_mean<<16 | flags<<24}
__device__ __forceinline__ uint4 pack_stats(uint64_t sumsq, uint32_t peak, uint32_t bsum, uint32_t n, bool alaw,
                                            bool probe, uint32_t &byte_mean, uint32_t &flags)
{
    byte_mean = (bsum / n) & 255u;
    flags = (peak <= 8u ? IGDSP_FLAG_SILENT : 0u) | (probe ? IGDSP_FLAG_PROBE_D5 : 0u) |
            (peak == full_scale(alaw) ? IGDSP_FLAG_CLIPPED : 0u);
    const float rms = sqrtf((float)sumsq / (float)n);
    return make_uint4((uint32_t)sumsq, (uint32_t)(sumsq >> 32), __float_as_uint(rms), peak | (byte_mean << 16) | (flags << 24));
}

__device__ __forceinline__ igdsp_frame_stats make_stats(uint64_t sumsq, uint32_t peak, uint32_t bsum, uint32_t n,
                                                        bool alaw, bool probe)
{
    igdsp_frame_stats st;
    st.sumsq = sumsq;
    st.rms = sqrtf((float)sumsq / (float)n);   // IEEE divide + sqrt (hipcc default: correctly rounded)
    st.peak = (uint16_t)peak;
    st.byte_mean = (uint8_t)(bsum / n);
    st.flags = (uint8_t)((peak <= 8u ? IGDSP_FLAG_SILENT : 0) | (probe ? IGDSP_FLAG_PROBE_D5 : 0) |
                         (peak == full_scale(alaw) ? IGDSP_FLAG_CLIPPED : 0));
    return st;
}

// Wave64 reductions on the VALU's DPP path (no LDS round trips): an inclusive scan inside each row of 16
// lanes (row_shr 1/2/4/8), then row_bcast15 / row_bcast31 carry the row totals upward; lane 63 ends up
// with the wave total and is read out with v_readlane.  Identity 0 suits unsigned add and max.
template <typename Op>
__device__ __forceinline__ uint32_t wave_reduce_dpp(uint32_t v, Op op)
{
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
struct OpAdd { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return max(a, b); } };

// Barrier-free launch-aggregate commit for the persistent kernels: a wave that has run out of work reduces its partials with DPP (no LDS, no shuffle tree), lane 0 adds the
// wave totals into a 48-byte block accumulator with seven LDS atomics and bumps a counter; the
// wave that finds the counter at nwaves - 1 moves the block totals to the aggregate with device atomics.  No wave waits for
// another, so the kernel's tail is one LDS round trip instead of two barriers and two shuffle trees (agg_commit_block):
// measured on k_meter_chunk64, 0.2314 -> 0.2294 ms per launch (the kernel without any aggregate: 0.2281).
struct AggBlock { unsigned long long sumsq, samples; uint32_t frames, n_silent, n_clipped, bm_sum, peak, done; };

__device__ __forceinline__ void agg_block_init(AggBlock &b)      // one thread, before a barrier every wave passes
{
    b.sumsq = 0; b.samples = 0; b.frames = 0; b.n_silent = 0; b.n_clipped = 0; b.bm_sum = 0; b.peak = 0; b.done = 0;
}

// sumsq / bm_sum / peak: per lane.  samples / frames / n_silent / n_clipped: wave-uniform (lane 0 adds them).
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)      // 24-bit limbs: 64 of them fit 32 bits
{
    const uint32_t s0 = wave_reduce_dpp((uint32_t)v & 0xFFFFFFu, OpAdd{});
    const uint32_t s1 = wave_reduce_dpp((uint32_t)(v >> 24) & 0xFFFFFFu, OpAdd{});
    const uint32_t s2 = wave_reduce_dpp((uint32_t)(v >> 48), OpAdd{});
    return (uint64_t)s0 + ((uint64_t)s1 << 24) + ((uint64_t)s2 << 48);
}

// sumsq / samples / bm_sum / peak: per lane.  frames / n_silent / n_clipped: wave-uniform.  When the block's last wave is
// out, the last BLOCK out re-arms the device work queue gq (nullptr: static schedule) for the next launch.
// agg == nullptr: only the counters move.
__device__ __forceinline__ void wave_exit(igdsp_aggregate *agg, uint32_t rank, AggBlock &b, uint32_t nwaves, uint32_t lane,
                                          uint32_t *gq, uint32_t G, uint64_t sumsq, uint64_t samples, uint32_t bm_sum, uint32_t peak,
                                          uint32_t frames, uint32_t n_silent, uint32_t n_clipped)
{
    // (all 64 lanes adding into one LDS address measured 6 us per block: same-address LDS atomics run one lane at a time)
    uint64_t tot = 0, smp = 0;
    uint32_t bm = 0, pk = 0;
    if (agg != nullptr) {
        tot = wave_sum_u64(sumsq);
        smp = wave_sum_u64(samples);
        bm = wave_reduce_dpp(bm_sum, OpAdd{});
        pk = wave_reduce_dpp(peak, OpMax{});
    }
    if (lane == 0) {
        if (agg != nullptr) {
            atomicAdd(&b.sumsq, (unsigned long long)tot);
            atomicAdd(&b.samples, (unsigned long long)smp);
            atomicAdd(&b.bm_sum, bm);
            atomicMax(&b.peak, pk);
            atomicAdd(&b.frames, frames);
            atomicAdd(&b.n_silent, n_silent);
            atomicAdd(&b.n_clipped, n_clipped);
        }
        // LDS operations of one wave execute in order, so every add above has landed when the counter moves
        if (__hip_atomic_fetch_add(&b.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == nwaves - 1u) {
            if (agg != nullptr && b.frames != 0u) {
                atomicAdd((unsigned long long *)&agg->sumsq, b.sumsq);
                atomicAdd((unsigned long long *)&agg->samples, b.samples);
                atomicAdd((unsigned long long *)&agg->frames, (unsigned long long)b.frames);
                atomicAdd((unsigned long long *)&agg->n_silent, (unsigned long long)b.n_silent);
                atomicAdd((unsigned long long *)&agg->n_clipped, (unsigned long long)b.n_clipped);
                atomicAdd((unsigned long long *)&agg->byte_mean_sum, (unsigned long long)b.bm_sum);
                atomicMax((unsigned long long *)&agg->peak_slot[rank & (IGDSP_AGG_MAX_RANKS - 1)], (unsigned long long)b.peak);
            }
            if (gq != nullptr && atomicAdd(gq + 1, 1u) == G - 1u) { gq[0] = 0u; gq[1] = 0u; }
        }
    }
}

// Merge one work item's window into hold[c] when several items share a channel (frame segments): device-scope integer
// atomics (adds, and a CAS loop on the {peak_hold, level_max, level_min} word) — exact and order-independent, so the
// result is bit-identical to the sequential fold (keeplogAudioLevel, Functions.cpp:2126-2145).
__device__ __forceinline__ void hold_merge(igdsp_chan_hold *g, const igdsp_chan_hold &h)
{
    atomicAdd((unsigned long long *)&g->sumsq_acc, (unsigned long long)h.sumsq_acc);
    atomicAdd(&g->count, h.count); atomicAdd(&g->level_sum, h.level_sum); atomicAdd(&g->samples, h.samples);
    atomicAdd(&g->n_silent, h.n_silent); atomicAdd(&g->n_clipped, h.n_clipped);
    uint32_t *pw = reinterpret_cast<uint32_t *>(&g->peak_hold);
    uint32_t old = *pw, want;
    do {
        const uint32_t pk = max(old & 0xFFFFu, (uint32_t)h.peak_hold), mx = max((old >> 16) & 0xFFu, (uint32_t)h.level_max);
        const uint32_t mn = min(old >> 24, (uint32_t)h.level_min);
        want = pk | (mx << 16) | (mn << 24);
        if (want == old) break;
        const uint32_t seen = atomicCAS(pw, old, want);
        if (seen == old) break;
        old = seen;
    } while (true);
}

// the same when ONE work item owns the channel for the whole launch: plain read-modify-write
__device__ __forceinline__ void hold_add(igdsp_chan_hold *gp, const igdsp_chan_hold &h)
{
    igdsp_chan_hold g = *gp;
    g.sumsq_acc += h.sumsq_acc; g.count += h.count; g.level_sum += h.level_sum; g.samples += h.samples;
    g.peak_hold = max(g.peak_hold, h.peak_hold); g.level_max = max(g.level_max, h.level_max); g.level_min = min(g.level_min, h.level_min);
    g.n_silent += h.n_silent; g.n_clipped += h.n_clipped;
    *gp = g;
}

// ============================================================================
// Shared machinery of the tuned n == 160 kernels (k_meter_chunk64, k_meter_rtp64, k_roundtrip_chunk64).
//
// Expansion LUT: 256 entries (law<<7 | code&0x7F) x 32 replicas x 8 B = 64 KiB
// in LDS, entry = { (|x|/4)^2 , |x| }.  Replica r sits at byte offset r*8 of the
// entry's 256-byte row, and lane l always reads replica l&31, so every
// ds_read_b64 of a 32-lane group touches 32 distinct 8-byte slots = all 64
// banks once: conflict-free for ANY code distribution.  The address is built
// by ONE v_perm_b32: byte0 = replica offset, byte1 = law|code7.
// (|x|/4)^2 <= 8064^2 < 2^26 so 16 samples fit a u32 partial; x^2 = 16 * that.
// ============================================================================
constexpr int kLutEntries = 256 * 32;   // uint2 each

__device__ __forceinline__ void fill_lut(uint2 *lut)
{
    for (uint32_t i = threadIdx.x; i < (uint32_t)kLutEntries; i += blockDim.x) {
        const uint32_t e = i >> 5;                 // law<<7 | code7
        const uint32_t ax = (e & 0x80u) ? alaw_abs(e) : ulaw_abs(e);
        const uint32_t m = ax >> 2;
        lut[i] = make_uint2(m * m, ax);
    }
}

__device__ __forceinline__ uint2 lut_at(const uint2 *lut, uint32_t t, uint32_t off, uint32_t sel)
{
    // byte address = off | (byte_k(t) << 8); v_perm_b32: sel bytes 4..7 pick from t, 0..3 from off, 0x0C = 0x00
    const uint32_t addr = __builtin_amdgcn_perm(t, off, sel);
    return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lut) + addr);
}

// The record-only kernels that are NOT bound by HBM (k_meter_strided: 6 % under the bare stream with its own access pattern,
// tools/stream_pieces.py) read a 4-byte LUT: entry = m * m alone (m = |x| >> 2, 26 bits).  A wave-wide ds_read_b32 moves 256 B
// through the 128 B/clk LDS port in 2 clocks where the ds_read_b64 of the 8-byte table takes 4, the lookup result needs one
// register instead of two, and NO per-sample masking is left: the row is the raw code byte (256 rows x 256 B, the v_perm address
// trick again; the sign bit just selects a duplicate), inside a row 32 replicas of the mu-law entry then 32 of the A-law entry,
// so the law is one bit of the per-piece offset instead of bit 7 of every code byte.  64 KiB for both laws.  The peak comes from
// the SAME value: max(m * m) is the square of max(m), and one float square root per 16-sample piece turns it back (isqrt_m2:
// exact, see there).  (k_meter_chunk64 IS bound by HBM: it measured 0.2278 ms with either table and keeps the 8-byte one, which
// the PCM variants need anyway.)
constexpr int kLut32Words = 256 * 64;
__device__ __forceinline__ void fill_lut32(uint32_t *lut)
{
    for (uint32_t i = threadIdx.x; i < (uint32_t)kLut32Words; i += blockDim.x) {
        const uint32_t e = ((i & 32u) << 2) | ((i >> 6) & 0x7Fu);   // law << 7 | code7
        const uint32_t m = ((e & 0x80u) ? alaw_abs(e) : ulaw_abs(e)) >> 2;
        lut[i] = m * m;
    }
}

__device__ __forceinline__ uint32_t lut32_at(const uint32_t *lut, uint32_t w, uint32_t off, uint32_t sel)
{
    // byte address = off | (byte_k(w) << 8), off = replica * 4 | law << 7; w holds raw code bytes
    const uint32_t addr = __builtin_amdgcn_perm(w, off, sel);
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lut) + addr);
}

// m from m * m (m <= 8064): (float)v is off by <= 2^-24 relative, v_sqrt_f32 by 1 ulp, so the root is within 0.002 of the
// integer m and rounding to nearest returns it exactly (every G.711 magnitude of both laws is checked through the record's peak
// in tests/test_gpu_parity.py).
__device__ __forceinline__ uint32_t isqrt_m2(uint32_t v)
{
    return (uint32_t)(__builtin_amdgcn_sqrtf((float)v) + 0.5f);
}

typedef short v2i16 __attribute__((ext_vector_type(2)));
typedef unsigned short v2u16_t __attribute__((ext_vector_type(2)));

// two magnitudes -> one dword of signed int16 PCM (codes k and k + 1 of word w; a code is negative iff its bit 7 is
// clear).  Packed 16-bit math: one v_perm puts the two inverted sign bits at bits 15 / 31, a packed arithmetic shift
// turns them into 0x0000 / 0xFFFF masks m, and (p ^ m) - m negates the selected halves: 5 VALU per PAIR.
__device__ __forceinline__ uint32_t pack_pcm(uint32_t w, uint32_t k, uint32_t ax0, uint32_t ax1)
{
    const uint32_t sb = __builtin_amdgcn_perm(~w, 0u, k == 0u ? 0x050C040Cu : 0x070C060Cu);
    const v2i16 m = __builtin_bit_cast(v2i16, sb) >> (v2i16)(15);
    const uint32_t p = ax0 | (ax1 << 16);
    const v2i16 r = __builtin_bit_cast(v2i16, p ^ __builtin_bit_cast(uint32_t, m)) - m;
    return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ void wave_lds_fence()
{
    // same-wave LDS hand-off (lane A writes, lane B reads): the LDS pipe is in
    // order per wave; this only stops the compiler from moving accesses across.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Silence-probe helper.  Bytes 28 / 38 / 48 of a frame sit in its pieces 1 / 2 / 3
// (byte 12 of piece 1, byte 6 of piece 2, byte 0 of piece 3).  Two constant-selector
// v_perm_b32 gather {d.w.b0, d.x.b0, d.y.b2} into one word; a per-lane mask keeps
// the byte this piece is responsible for.  Returns bit 31 set when the probe fails.
__device__ __forceinline__ uint32_t probe_fail(const uint4 d, const uint32_t pmask)
{
    const uint32_t y = __builtin_amdgcn_perm(d.w, d.x, 0x0C0C0004u);      // b0 = d.w.b0 (byte 28), b1 = d.x.b0 (byte 48)
    const uint32_t x = __builtin_amdgcn_perm(d.y, y, 0x0C060100u);        // b2 = d.y.b2 (byte 38)
    return min((x ^ 0x00D5D5D5u) & pmask, 1u) << 31;
}

__device__ __forceinline__ uint32_t probe_mask(uint32_t q)   // q = piece index within the frame
{
    return (q == 1u ? 0x000000FFu : 0u) | (q == 3u ? 0x0000FF00u : 0u) | (q == 2u ? 0x00FF0000u : 0u);
}

__device__ __forceinline__ uint64_t now_cycles() { return __builtin_readcyclecounter(); }

// Frame record for the tuned n == 160 path.  sumsq = 16 * s with s < 2^34; rms = sqrt(s / 10):
// two u32->f32 converts + one fma, one multiply, one v_sqrt_f32 (1 ulp) — total relative error
// < 4e-7 against the float64 definition, inside the 1e-5 contract; every integer field is exact.
__device__ __forceinline__ uint4 pack_stats160(uint64_t s, uint32_t peak, uint32_t bsum, bool alaw, bool probe,
                                               uint32_t &byte_mean, uint32_t &flags)
{
    byte_mean = bsum / 160u;
    flags = (peak <= 8u ? IGDSP_FLAG_SILENT : 0u) | (probe ? IGDSP_FLAG_PROBE_D5 : 0u) |
            (peak == full_scale(alaw) ? IGDSP_FLAG_CLIPPED : 0u);
    const float fs = fmaf((float)(uint32_t)(s >> 32), 4294967296.0f, (float)(uint32_t)s);
    const float rms = __builtin_amdgcn_sqrtf(fs * 0.1f);
    const uint64_t sumsq = s << 4;
    return make_uint4((uint32_t)sumsq, (uint32_t)(sumsq >> 32), __float_as_uint(rms), peak | (byte_mean << 16) | (flags << 24));
}

constexpr int kSuperFrames = 2 * kChunkFrames;                 // 64
constexpr int kStripEntries = kSuperFrames * kPiecesPerFrame;  // 640 x 8 B = 5 KiB per wave
// Order in which batches of work items are visited: the two halves of the item range alternately, so that at any
// moment the launch reads and WRITES in two distant places of every buffer.  Write streams spread over two classes of
// device memory run 11-22 % faster on MI355X than the same stream into one class (tools/stream_calib2.py, DESIGN.md
// 7); a caller gets that by letting a bulk output buffer straddle a class boundary.  A bijection on [0, nb); ids >=
// nb (queue exhausted) are returned unchanged.
#ifndef IGDSP_SPREAD_METER
#define IGDSP_SPREAD_METER 0
#endif
__device__ __forceinline__ uint32_t spread_batch(uint32_t b, uint32_t nb)
{
#ifdef IGDSP_NO_SPREAD          // A/B builds only: ascending order everywhere
    return b;
#endif
    const uint32_t half = (nb + 1u) >> 1;
    return b >= nb ? b : ((b & 1u) ? half + (b >> 1) : (b >> 1));
}

// Work queue of a persistent block (the mechanism k_meter_chunk64 carries inline): a batch = W consecutive items; the
// block's first batch is its blockIdx, later ones come from one device-wide counter (gq[0]; nullptr = static
// blockIdx + j * G); the waves draw slots from an LDS counter and the wave that draws the first slot of local batch j
// announces batch j + 1, so nobody waits on the device atomic.  gq[1] counts finished blocks; the last one re-arms.
template <int W>
struct BlockQueue { uint32_t next, batch[8], tag[8]; };

template <int W>
__device__ __forceinline__ void bq_init(BlockQueue<W> &q, uint32_t *gq, uint32_t G, uint32_t gb1)   // thread 0, before a barrier;
{                                                                 // gb1 = atomicAdd(gq, 1u) issued earlier (its latency hides under the LUT fill)
    q.next = (uint32_t)W;
    for (int i = 0; i < 8; ++i) q.tag[i] = 0xFFFFFFFFu;
    q.batch[0] = blockIdx.x; q.tag[0] = 0u;
    q.batch[1] = gq ? gb1 + G : blockIdx.x + G; q.tag[1] = 1u;
}

template <int W>
__device__ __forceinline__ uint32_t bq_grab(BlockQueue<W> &q, uint32_t *gq, uint32_t G, uint32_t lane, uint32_t nb)   // wave-uniform item id; nb = number of batches
{
    uint32_t v = 0;
    if (lane == 0) {
        const uint32_t s = atomicAdd(&q.next, 1u);
        const uint32_t j = s / (uint32_t)W, w = s - j * (uint32_t)W;
        if (w == 0u) {
            const uint32_t nb = gq ? atomicAdd(gq, 1u) + G : blockIdx.x + (j + 1u) * G;
            __hip_atomic_store(&q.batch[(j + 1u) & 7u], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&q.tag[(j + 1u) & 7u], j + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        while (__hip_atomic_load(&q.tag[j & 7u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != j)
            __builtin_amdgcn_s_sleep(2);                         // published by a wave of this block that never waits on us
        v = spread_batch(__hip_atomic_load(&q.batch[j & 7u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), nb) * (uint32_t)W + w;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ void bq_finish(uint32_t *gq, uint32_t G)                     // all threads, end of the kernel
{
    if (gq == nullptr) return;
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(gq + 1, 1u) == G - 1u) { gq[0] = 0u; gq[1] = 0u; }
}


}  // namespace igdsp
