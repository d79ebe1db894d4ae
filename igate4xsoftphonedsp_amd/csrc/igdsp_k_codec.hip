// igdsp_k_codec.hip — G.711 compression (k_encode_*), the fused round trip (k_roundtrip_*), the hold / window kernels.
// Hand-written gfx950 (CDNA4, wave64) kernels; no MFMA: the path is a byte stream with ~4 integer ops per sample, bounded by
// HBM.  Shared device code: igdsp_device.h.
#include "igdsp_device.h"

namespace igdsp {

// ============================================================================
// a2 — G.711 compression.  ONE branch-free formulation serves both laws and both encoder lineages
// (include/igdsp.h): per-law constants select bias / rounding, the segment comes from count-leading-
// zeros.  With msb = 31 - clz(mag):
//   SUN16  mu : mag = min(|v| + 0x84, 0x7FFF)                 A : mag = v >= 0 ? v : max(-v - 8, 0)
//          seg = max(msb,7) - 7        step = (mag >> (max(msb, mu?7:8) - 4)) & 15
//   G191   mu : mag = min(|v>>2| + 0x21, 0x1FFF)              A : mag = (v>>3) ^ sign   (= -x-1 for x < 0)
//          seg = max(msb, mu?5:4) - (mu?5:4)   step = (mag >> (max(msb,5) - 4)) & 15
//   code = (seg<<4 | step) ^ (mu ? 0xFF : 0xD5) ^ (v < 0 ? 0x80 : 0)
// (clamping mag is identical to the classic "segment 8 -> 0x7F ^ mask" overflow rule).
// ============================================================================
struct EncK { int k_and, k_add, sh; uint32_t c_shift, c_seg, base; };

template <int VARIANT>
__device__ __forceinline__ EncK enc_consts(bool alaw)
{
    EncK k;
    if (VARIANT == IGDSP_ENC_SUN16) {
        k.k_and = alaw ? -8 : 0; k.k_add = alaw ? 0 : 0x84; k.sh = 0;
        k.c_shift = alaw ? 23u : 24u;      // 31 - floor(msb) for the step shift
        k.c_seg = 24u;                     // 31 - 7
    } else {
        k.k_and = alaw ? 0 : 1; k.k_add = alaw ? 0 : 0x21; k.sh = alaw ? 3 : 2;
        k.c_shift = 26u;                   // 31 - 5
        k.c_seg = alaw ? 27u : 26u;        // 31 - {4,5}
    }
    k.base = alaw ? 0xD5u : 0xFFu;
    return k;
}

template <int VARIANT>
__device__ __forceinline__ uint32_t enc_uni(int v, const EncK k)
{
    int mag;
    const int sign = v >> 31;                                   // -1 for negative samples
    if (VARIANT == IGDSP_ENC_SUN16) {
        const int av = (v ^ sign) - sign;                       // |v|, 32768 for -32768
        mag = min(max(av + ((sign & k.k_and) + k.k_add), 0), 0x7FFF);
    } else {
        const int vd = v >> k.sh;                               // arithmetic: floors negatives
        const int t = vd ^ sign;                                // x >= 0 ? x : -x - 1
        mag = min(t + (sign & k.k_and) + k.k_add, 0x1FFF);      // mu: |x| + 0x21 ; A: -x - 1
    }
    const uint32_t c = (uint32_t)__clz(mag);                    // 32 for mag == 0
    const uint32_t shift = 27u - min(c, k.c_shift);             // max(msb, floor) - 4
    const uint32_t sg = k.c_seg - min(c, k.c_seg);              // max(msb, f) - f
    const uint32_t step = ((uint32_t)mag >> shift) & 15u;
    return ((sg << 4) | step) ^ k.base ^ ((uint32_t)sign & 0x80u);
}

// ----------------------------------------------------------------------------
// Table-driven form of the same compressor (what production G.711 encoders do): 2 laws x 16 384 cells of
// one byte in LDS, a cell = four neighbouring PCM values on which the compressor is constant:
//   SUN16 works on sign / magnitude      -> cell = (v < 0, |v| >> 2)         (|v| >> 2 clamped to 8191)
//   G191  works on the floored 14-bit value -> cell = (v >> 2) + 8192
// The table is generated at kernel start by running enc_uni on one representative value per cell.
// ----------------------------------------------------------------------------
constexpr int kEncCells = 16384;

template <int VARIANT>
__device__ __forceinline__ uint32_t enc_cell(int v)
{
    if (VARIANT == IGDSP_ENC_SUN16) {
        const int sign = v >> 31;
        const int av = (v ^ sign) - sign;
        return ((uint32_t)sign & 8192u) + min((uint32_t)av >> 2, 8191u);
    }
    return (uint32_t)((v >> 2) + 8192);
}

template <int VARIANT>
__device__ __forceinline__ int enc_cell_value(uint32_t cell)
{
    if (VARIANT == IGDSP_ENC_SUN16) {
        const int k = (int)(cell & 8191u);
        return (cell & 8192u) ? -(4 * k + 1) : 4 * k;          // (neg, k = 0) is {-1,-2,-3}: zero is never negative
    }
    return ((int)cell - 8192) * 4;
}

template <int VARIANT>
__device__ __forceinline__ void fill_enc_table(uint8_t *tab)     // tab[2][kEncCells]: mu-law, A-law
{
    const EncK ku = enc_consts<VARIANT>(false), ka = enc_consts<VARIANT>(true);
    for (uint32_t i = threadIdx.x; i < (uint32_t)kEncCells; i += blockDim.x) {
        const int v = enc_cell_value<VARIANT>(i);
        tab[i] = (uint8_t)enc_uni<VARIANT>(v, ku);
        tab[kEncCells + i] = (uint8_t)enc_uni<VARIANT>(v, ka);
    }
}

// Diagnostic/test entry: the table-driven compressor on arbitrary PCM (exhaustive parity test of the cells).
template <int VARIANT>
__global__ __launch_bounds__(1024) void k_encode_table(const int16_t *__restrict__ pcm, const uint8_t *__restrict__ codec,
                                                       uint32_t C, uint32_t n, uint64_t n_samples, uint8_t *__restrict__ out)
{
    __shared__ uint8_t tab[2 * kEncCells];
    fill_enc_table<VARIANT>(tab);
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)((i / n) % C);
        out[i] = tab[(codec[c] == IGDSP_PT_PCMA ? kEncCells : 0) + enc_cell<VARIANT>((int)pcm[i])];
    }
}

// 8 samples (16 B) per lane in, 8 codes (8 B) out; requires n % 8 == 0 and 16 B aligned pcm.
template <int VARIANT>
__global__ __launch_bounds__(256) void k_encode_v8(const int16_t *__restrict__ pcm, const uint8_t *__restrict__ codec,
                                                   uint32_t C, uint32_t n, uint64_t n_groups, uint8_t *__restrict__ out)
{
    const uint32_t groups_per_frame = n >> 3;
    auto encode_group = [&](uint64_t g, const uint4 d) {
        const uint32_t c = (uint32_t)((g / groups_per_frame) % C);
        const EncK k = enc_consts<VARIANT>(codec[c] == IGDSP_PT_PCMA);
        const uint32_t w[4] = {d.x, d.y, d.z, d.w};
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[2 * i] = enc_uni<VARIANT>((int)(int16_t)(w[i] & 0xFFFFu), k);
            r[2 * i + 1] = enc_uni<VARIANT>((int)(int16_t)(w[i] >> 16), k);
        }
        uint2 o;
        o.x = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        o.y = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
        reinterpret_cast<uint2 *>(out)[g] = o;
    };
    // four independent 16-byte loads in flight per lane (one per quarter of the grid-stride step)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 *src = reinterpret_cast<const uint4 *>(pcm);
    for (; g + 3u * stride < n_groups; g += 4u * stride) {
        const uint4 d0 = ld_stream(src + g), d1 = ld_stream(src + g + stride), d2 = ld_stream(src + g + 2u * stride),
                    d3 = ld_stream(src + g + 3u * stride);
        encode_group(g, d0); encode_group(g + stride, d1); encode_group(g + 2u * stride, d2); encode_group(g + 3u * stride, d3);
    }
    for (; g < n_groups; g += stride) encode_group(g, ld_stream(src + g));
}

// Same 8-samples-per-lane geometry with the table-driven compressor (2 x 16 384 one-byte cells in LDS, built per
// block by enc_uni): ~8 VALU + one LDS byte read per sample instead of ~20 VALU, which moves the encode kernel from
// VALU-bound towards the copy-like HBM bound.  Persistent blocks so the 32 KiB table is built 2 x CUs times only.
template <int VARIANT>
__global__ __launch_bounds__(1024) void k_encode_v8_table(const int16_t *__restrict__ pcm, const uint8_t *__restrict__ codec,
                                                          uint32_t C, uint32_t n, uint64_t n_groups, uint8_t *__restrict__ out)
{
    __shared__ uint8_t tab[2 * kEncCells];
    fill_enc_table<VARIANT>(tab);
    __syncthreads();
    const uint32_t groups_per_frame = n >> 3;
    auto encode_group = [&](uint64_t g, const uint4 d) {
        const uint32_t c = (uint32_t)((g / groups_per_frame) % C);
        const uint8_t *t = tab + (codec[c] == IGDSP_PT_PCMA ? kEncCells : 0);
        const uint32_t w[4] = {d.x, d.y, d.z, d.w};
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[2 * i] = t[enc_cell<VARIANT>((int)(int16_t)(w[i] & 0xFFFFu))];
            r[2 * i + 1] = t[enc_cell<VARIANT>((int)(int16_t)(w[i] >> 16))];
        }
        uint2 o;
        o.x = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        o.y = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
        reinterpret_cast<uint2 *>(out)[g] = o;
    };
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 *src = reinterpret_cast<const uint4 *>(pcm);
    for (; g + 3u * stride < n_groups; g += 4u * stride) {
        const uint4 d0 = ld_stream(src + g), d1 = ld_stream(src + g + stride), d2 = ld_stream(src + g + 2u * stride),
                    d3 = ld_stream(src + g + 3u * stride);
        encode_group(g, d0); encode_group(g + stride, d1); encode_group(g + 2u * stride, d2); encode_group(g + 3u * stride, d3);
    }
    for (; g < n_groups; g += stride) encode_group(g, ld_stream(src + g));
}

// Large batches: the compressor as a FULL 16-bit table, tab[law][uint16(v)] = 128 KiB of LDS built per block by
// enc_uni (one block per CU, kEncWaves waves).  The whole LDS address {law, v.hi, v.lo} is one v_perm_b32 of the loaded
// PCM word, so a sample costs ~2 VALU + one ds_read_u8 and the kernel sits on the copy-like HBM bound.  Frame /
// channel bookkeeping is incremental (adds and compares): the grid-stride step is decomposed once per thread into
// whole frames + groups, so no division runs inside the loop.
// Waves per block (late round 3, same-box A/B builds): 16 / 14 / 12 / 10 / 8 / 6 / 4 → 0.651–0.662 / 0.650–0.658 / 0.641–0.642 / 0.636–0.637 / 0.646–0.648 /
// 0.770 / 1.05 ms.  The 2 : 1 read : write mix sits between the read-heavy kernels (the more waves the better) and the store / round-trip
// kernels (4–6): ten waves of 8 KiB chunks keep enough loads in flight, more only add write fronts.
#ifndef IGDSP_ENC_WAVES
#define IGDSP_ENC_WAVES 10
#endif
constexpr int kEncWaves = IGDSP_ENC_WAVES;
template <int VARIANT>
__global__ __launch_bounds__(kEncWaves * 64) void k_encode_lut16(const int16_t *__restrict__ pcm, const uint8_t *__restrict__ codec,
                                                       uint32_t C, uint32_t n, uint32_t n_groups, uint8_t *__restrict__ out,
                                                       uint32_t *gqueue, const uint8_t *__restrict__ tab_g)
{
    constexpr int kW = kEncWaves;
    __shared__ __attribute__((aligned(16))) uint8_t tab[2 * 65536];
    __shared__ BlockQueue<kW> bq;
    uint32_t gb1 = 0;
    if (threadIdx.x == 0 && gqueue != nullptr) gb1 = atomicAdd(gqueue, 1u);
    if (tab_g != nullptr) {                                       // the context's ready-made table (k_build_enc_table): 128 KiB out of L2
        for (uint32_t i = threadIdx.x * 16u; i < 2u * 65536u; i += blockDim.x * 16u)
            *reinterpret_cast<uint4 *>(tab + i) = *reinterpret_cast<const uint4 *>(tab_g + i);
    } else {
        const EncK ku = enc_consts<VARIANT>(false), ka = enc_consts<VARIANT>(true);
        for (uint32_t i = threadIdx.x; i < 2u * 65536u; i += blockDim.x)
            tab[i] = (uint8_t)enc_uni<VARIANT>((int)(int16_t)(i & 0xFFFFu), (i >> 16) ? ka : ku);
    }
    if (threadIdx.x == 0) bq_init(bq, gqueue, gridDim.x, gb1);
    __syncthreads();
    const uint32_t gpf = n >> 3;                                   // 8-sample groups per frame
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, G = gridDim.x;
    auto encode_group = [&](uint32_t gi, const uint4 d, uint32_t pt) {
        const uint32_t law = pt == IGDSP_PT_PCMA ? 1u : 0u;        // becomes address byte 2: +64 KiB
        const uint32_t w[4] = {d.x, d.y, d.z, d.w};
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[2 * i] = tab[__builtin_amdgcn_perm(w[i], law, 0x0C000504u)];
            r[2 * i + 1] = tab[__builtin_amdgcn_perm(w[i], law, 0x0C000706u)];
        }
        uint2 o;
        o.x = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        o.y = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
        reinterpret_cast<uint2 *>(out)[gi] = o;
    };
    const uint4 *src = reinterpret_cast<const uint4 *>(pcm);
    // A wave takes 8 KiB chunks (kP pieces of 64 x 16 B, contiguous) from the block / device work queue (batches of 16
    // neighbouring chunks); piece j's register is reloaded from the wave's next chunk as soon as piece j is encoded.
    constexpr int kP = 8;
    constexpr uint32_t kChunkGroups = 64u * kP;
    const uint32_t n_chunks = n_groups / kChunkGroups;
    // channel of piece j of a chunk = channel of its piece 0 advanced by 64 j groups: wave-uniform (frames, groups) steps
    uint32_t d_r[kP], d_c[kP];
#pragma unroll
    for (int j = 0; j < kP; ++j) { d_r[j] = (64u * j) % gpf; d_c[j] = ((64u * j) / gpf) % C; }
    auto channels_of = [&](uint32_t chunk, uint32_t (&cc)[kP]) {
        const uint32_t g0 = chunk * kChunkGroups + lane, f = g0 / gpf, gin0 = g0 - f * gpf, c0 = f % C;
#pragma unroll
        for (int j = 0; j < kP; ++j) {
            uint32_t gi = gin0 + d_r[j], c = c0 + d_c[j];
            if (gi >= gpf) c += 1u;
            if (c >= C) c -= C;
            cc[j] = c;
        }
    };
    const uint32_t n_batches = (n_chunks + (uint32_t)kW - 1u) / (uint32_t)kW;
    uint32_t chunk = spread_batch(blockIdx.x, n_batches) * (uint32_t)kW + wave;
    if (chunk < n_chunks) {
        uint4 d[kP];
        uint32_t pt[kP], cc[kP];
        channels_of(chunk, cc);
#pragma unroll
        for (int j = 0; j < kP; ++j) { d[j] = ld_stream(src + (chunk * kChunkGroups + lane + 64u * j)); pt[j] = codec[cc[j]]; }
        uint32_t next = bq_grab(bq, gqueue, G, lane, n_batches);
        for (;;) {
            const bool has_next = next < n_chunks;
            const uint32_t nl = has_next ? next : chunk;           // last round re-reads itself: loads stay unconditional
            const uint32_t gl = nl * kChunkGroups + lane, gs = chunk * kChunkGroups + lane;
            channels_of(nl, cc);
#pragma unroll
            for (int j = 0; j < kP; ++j) {
                encode_group(gs + 64u * j, d[j], pt[j]);
                d[j] = ld_stream(src + (gl + 64u * j));
                pt[j] = codec[cc[j]];
            }
            if (!has_next) break;
            chunk = next;
            next = bq_grab(bq, gqueue, G, lane, n_batches);
        }
    }
    // groups beyond the last whole chunk (< 512): plain grid-stride
    for (uint32_t g = n_chunks * kChunkGroups + blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x)
        encode_group(g, src[g], codec[(g / gpf) % C]);
    bq_finish(gqueue, G);
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_encode_scalar(const int16_t *__restrict__ pcm, const uint8_t *__restrict__ codec,
                                                       uint32_t C, uint32_t n, uint64_t n_samples, uint8_t *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)((i / n) % C);
        out[i] = (uint8_t)enc_uni<VARIANT>((int)pcm[i], enc_consts<VARIANT>(codec[c] == IGDSP_PT_PCMA));
    }
}

// ============================================================================
// Config #5 — fused decode -> stats -> re-encode -> per-channel hold (a1 + a2 + a5 + a6).
// Channel-group-major: one wavefront owns 64 consecutive CHANNELS and walks all F frames of them
// (frame f of those channels is one contiguous 10 240-byte super-chunk at stride C*160), so the hold
// state — keeplogAudioLevel's count / sum / max / min (Functions.cpp:2126-2145) plus peak-hold and
// sum of squares — lives in the frame lanes' registers for the whole launch and is written once.
// Same LUT, strip and load pipeline as k_meter_chunk64; the re-encode is the full compression
// arithmetic (enc_uni) applied to the decoded PCM value, not a shortcut.  Needs C % 64 == 0, n == 160.
// ============================================================================
#ifndef IGDSP_RT_WAVES
#define IGDSP_RT_WAVES 12
#endif
constexpr int kRtWaves = IGDSP_RT_WAVES;
// LDS map of k_roundtrip_chunk64 (158 KiB of the CU's 160 KiB), chosen so that both table addresses come out
// of the instruction stream without adds:
//   [  0,  32 KiB)  compressor grid table  tab[law][neg][k] = enc(neg ? -4k : 4k), one byte per cell
//   [ 32,  62 KiB)  strips of waves 0..5
//   [ 64, 128 KiB)  expansion LUT; entry = {(|x|/4)^2, slot of the compressor cell of (law, |x|/4)}
//   [128, 158 KiB)  strips of waves 6..11
// Every G.711 expander output is a multiple of 4 with |x| <= 32256, so (law, sign, |x| / 4) enumerates the
// compressor's whole input domain on this path; the table holds the compressor (enc_uni, the same arithmetic
// igdsp_encode runs) evaluated at exactly those PCM values.  -0 (mu-law code 0x7F) lands in cell (neg, 0),
// which holds enc(0), as two's-complement PCM would.
constexpr uint32_t kRtEncBytes = 32768u, kRtLutOff = 65536u, kRtStripBytes = (uint32_t)kStripEntries * 8u;
constexpr uint32_t kRtStripA = kRtEncBytes, kRtStripB = kRtLutOff + (uint32_t)kLutEntries * 8u;
constexpr uint32_t kRtLdsBytes = kRtStripB + (uint32_t)(kRtWaves - kRtWaves / 2) * kRtStripBytes;
static_assert(kRtStripA + (uint32_t)(kRtWaves / 2) * kRtStripBytes <= kRtLutOff, "strips A overlap the LUT");
static_assert(kRtLdsBytes <= 160u * 1024u, "LDS budget");

// Slot of compressor cell t = law << 14 | k inside its 16 KiB half-table: t ^ (t >> 4).  G.711 expander outputs of the
// upper segments differ only in high bits of k (k = (2m + 33) * 2^s - 33), so with slot = k every mantissa of a segment
// lands in the SAME LDS bank (PMC: 66 % of the LDS cycles of this kernel were bank conflicts); folding the high bits into
// the bank bits takes the average cost of a 32-lane byte read from 3.9 to 2.5 cycles on D-speech.  A bijection on 15 bits
// that never touches bit 13 (the sign is OR-ed in afterwards).
__device__ __forceinline__ uint32_t rt_cell_slot(uint32_t t) { return t ^ (t >> 4); }

template <int VARIANT>
__device__ __forceinline__ void fill_rt_tables(uint8_t *smem)
{
    const EncK ku = enc_consts<VARIANT>(false), ka = enc_consts<VARIANT>(true);
    for (uint32_t i = threadIdx.x; i < kRtEncBytes; i += blockDim.x) {
        const int k = (int)(i & 8191u);
        const int v = (i & 8192u) ? -4 * k : 4 * k;
        smem[rt_cell_slot(i & ~8192u) | (i & 8192u)] = (uint8_t)enc_uni<VARIANT>(v, (i & 16384u) ? ka : ku);
    }
    uint2 *lut = reinterpret_cast<uint2 *>(smem + kRtLutOff);
    for (uint32_t i = threadIdx.x; i < (uint32_t)kLutEntries; i += blockDim.x) {
        const uint32_t e = i >> 5;                 // law<<7 | code7
        const uint32_t ax = (e & 0x80u) ? alaw_abs(e) : ulaw_abs(e);
        const uint32_t m = ax >> 2;
        lut[i] = make_uint2(m * m, rt_cell_slot(((e & 0x80u) << 7) | m));   // {(|x|/4)^2, slot of compressor cell (law, |x|/4)}
    }
}

// One half (32 frames) of a super-chunk.  Three-stage software pipeline per unit of 8 samples:
//   expansion-LUT reads of unit u+1 in flight | unit u folded, its 8 compressor-cell reads issued | unit u-1 packed
// `offx` = replica offset | 0x100: the 0x01 in byte 1 becomes address byte 2 (the LUT's 64 KiB base) inside the v_perm.
template <int VARIANT>
__device__ __forceinline__ void roundtrip_half(const uint8_t *smem, uint2 *strip_half, uint4 (&d)[kLoadsPerChunk],
                                               const uint32_t (&lm)[kLoadsPerChunk], const uint32_t (&pm)[kLoadsPerChunk],
                                               const uint32_t offx, const uint32_t lane, uint4 *out_half, const uint4 *refill)
{
    uint2 e[2][8];
    uint32_t wa[2], wb[2], eb[2][8];
    auto lut = [&](uint32_t t, uint32_t sel) {
        return *reinterpret_cast<const uint2 *>(smem + __builtin_amdgcn_perm(t, offx, sel));
    };
    auto issue = [&](int u) {
        const int j = u >> 1, k = u & 1;
        wa[k] = (u & 1) ? d[j].z : d[j].x;
        wb[k] = (u & 1) ? d[j].w : d[j].y;
        const uint32_t ta = (wa[k] & 0x7F7F7F7Fu) | lm[j], tb = (wb[k] & 0x7F7F7F7Fu) | lm[j];
        e[k][0] = lut(ta, 0x0C010400u); e[k][1] = lut(ta, 0x0C010500u); e[k][2] = lut(ta, 0x0C010600u); e[k][3] = lut(ta, 0x0C010700u);
        e[k][4] = lut(tb, 0x0C010400u); e[k][5] = lut(tb, 0x0C010500u); e[k][6] = lut(tb, 0x0C010600u); e[k][7] = lut(tb, 0x0C010700u);
    };
    // compressor cell of sample i of word w: its slot comes straight from entry.y, the sign from the code
    auto cells = [&](uint32_t w, const uint2 &e0, const uint2 &e1, const uint2 &e2, const uint2 &e3, uint32_t *dst) {
        const uint32_t nw = ~w & 0x80808080u;                    // bit 8i+7 set: sample i is negative
        dst[0] = smem[((nw << 6) & 0x2000u) | e0.y];
        dst[1] = smem[((nw >> 2) & 0x2000u) | e1.y];
        dst[2] = smem[((nw >> 10) & 0x2000u) | e2.y];
        dst[3] = smem[((nw >> 18) & 0x2000u) | e3.y];
    };
    auto pack = [&](const uint32_t *b) { return b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24); };
    uint32_t sum = 0, peak = 0, bsum = 0;
    uint32_t o[4];
    issue(0);
#pragma unroll
    for (int u = 0; u <= 2 * kLoadsPerChunk; ++u) {
        const int j = u >> 1, k = u & 1;
        if (u + 1 < 2 * kLoadsPerChunk) issue(u + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (u < 2 * kLoadsPerChunk) {
            bsum = __builtin_amdgcn_sad_u8(wa[k], 0u, bsum);
            bsum = __builtin_amdgcn_sad_u8(wb[k], 0u, bsum);
            sum = sum + e[k][0].x + e[k][1].x; sum = sum + e[k][2].x + e[k][3].x;
            sum = sum + e[k][4].x + e[k][5].x; sum = sum + e[k][6].x + e[k][7].x;
            peak = max(max(peak, e[k][0].x), e[k][1].x); peak = max(max(peak, e[k][2].x), e[k][3].x);   // max of (|x|/4)^2
            peak = max(max(peak, e[k][4].x), e[k][5].x); peak = max(max(peak, e[k][6].x), e[k][7].x);
            cells(wa[k], e[k][0], e[k][1], e[k][2], e[k][3], &eb[k][0]);
            cells(wb[k], e[k][4], e[k][5], e[k][6], e[k][7], &eb[k][4]);
            if (k == 1) {
                // peak |x| = 4 * sqrt(max (|x|/4)^2), (|x|/4) <= 8064
                const uint32_t pk = (uint32_t)(__builtin_amdgcn_sqrtf((float)peak) + 0.5f) << 2;   // (float)peak and v_sqrt_f32 are each within 1 ulp: round, never truncate
                strip_half[j * 64 + lane] = make_uint2(sum, pk | (bsum << 16) | probe_fail(d[j], pm[j]));
                d[j] = ld_stream(refill + j * 64);
                sum = 0; peak = 0; bsum = 0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u >= 1) {                                            // unit u-1: its cell bytes have had a whole fold to arrive
            const int jp = (u - 1) >> 1, kp = (u - 1) & 1;
            o[2 * kp] = pack(&eb[kp][0]);
            o[2 * kp + 1] = pack(&eb[kp][4]);
            if (kp == 1) st_stream(out_half + jp * 64, make_uint4(o[0], o[1], o[2], o[3]));
        }
    }
}

template <int VARIANT>
__global__ __launch_bounds__(kRtWaves * 64) void k_roundtrip_chunk64(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t F,
    uint8_t *__restrict__ out, igdsp_frame_stats *__restrict__ stats, igdsp_chan_hold *__restrict__ hold,
    const uint8_t *__restrict__ gate, uint32_t n_seg, uint32_t n_groups)
{
    // Work item = (channel group of 64, frame segment): with 65 536 channels there are only 1 024 groups, so the
    // launcher splits the F frames into n_seg segments to fill the chip.  n_seg == 1: the wave owns its channels'
    // hold records outright (plain read-modify-write).  n_seg > 1: each item folds its own window and MERGES it into
    // hold[c] with device-scope integer atomics (adds, and a CAS loop on the {peak_hold, max, min} word) —
    // exact and order-independent, so the result is bit-identical to the sequential fold.
    __shared__ __attribute__((aligned(16))) uint8_t smem[kRtLdsBytes];
    fill_rt_tables<VARIANT>(smem);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint2 *strip = reinterpret_cast<uint2 *>(smem + (wave < (uint32_t)(kRtWaves / 2) ? kRtStripA + wave * kRtStripBytes
                                                                                     : kRtStripB + (wave - (uint32_t)(kRtWaves / 2)) * kRtStripBytes));
    const uint32_t offx = (lane & 31u) * 8u | 0x100u;
    uint32_t fr[kLoadsPerChunk], pm[kLoadsPerChunk];
#pragma unroll
    for (int j = 0; j < kLoadsPerChunk; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane;
        fr[j] = p / 10u;
        pm[j] = probe_mask(p - fr[j] * 10u);
    }
    const uint32_t total_waves = gridDim.x * kRtWaves;
    const uint32_t fstride16 = C * (uint32_t)kPiecesPerFrame;       // uint4 units between frames of one channel group

    for (uint32_t item = wave * gridDim.x + blockIdx.x; item < n_groups * n_seg; item += total_waves) {
        const uint32_t seg = item / n_groups, cg = item - seg * n_groups;
        const uint32_t f_lo = (uint32_t)(((uint64_t)F * seg) / n_seg), f_hi = (uint32_t)(((uint64_t)F * (seg + 1u)) / n_seg);
        if (f_lo >= f_hi) continue;
        const uint32_t c0 = cg * kSuperFrames, cme = c0 + lane;
        const bool my_alaw = codec[cme] == IGDSP_PT_PCMA;
        const bool open = (gate == nullptr) || (gate[cme] != 0);
        igdsp_chan_hold h;
        if (n_seg == 1u) h = hold[cme];
        else { h.sumsq_acc = 0; h.count = 0; h.level_sum = 0; h.samples = 0; h.peak_hold = 0; h.level_max = 0; h.level_min = 255; h.n_silent = 0; h.n_clipped = 0; }
        const uint64_t amask = __ballot(my_alaw);
        const uint32_t am_lo = (uint32_t)amask, am_hi = (uint32_t)(amask >> 32);
        uint32_t lm0[kLoadsPerChunk], lm1[kLoadsPerChunk];
#pragma unroll
        for (int j = 0; j < kLoadsPerChunk; ++j) {
            lm0[j] = (uint32_t)__builtin_amdgcn_sbfe(am_lo, fr[j], 1) & 0x80808080u;
            lm1[j] = (uint32_t)__builtin_amdgcn_sbfe(am_hi, fr[j], 1) & 0x80808080u;
        }
        const uint4 *src = reinterpret_cast<const uint4 *>(payload) + ((uint64_t)c0 * kPiecesPerFrame + lane);
        uint4 *dst = reinterpret_cast<uint4 *>(out) + ((uint64_t)c0 * kPiecesPerFrame + lane);

        uint4 X[kLoadsPerChunk], Y[kLoadsPerChunk];
#pragma unroll
        for (int j = 0; j < kLoadsPerChunk; ++j) X[j] = ld_stream(src + (uint64_t)f_lo * fstride16 + j * 64);
#pragma unroll
        for (int j = 0; j < kLoadsPerChunk; ++j) Y[j] = ld_stream(src + (uint64_t)f_lo * fstride16 + kPiecesPerChunk + j * 64);

        for (uint32_t f = f_lo; f < f_hi; ++f) {
            const bool more = f + 1u < f_hi;                    // wave-uniform; the last frame re-reads itself (cache hit)
            const uint4 *nsrc = src + (uint64_t)(more ? f + 1u : f) * fstride16;
            uint4 *o16 = dst + (uint64_t)f * fstride16;
            roundtrip_half<VARIANT>(smem, strip, X, lm0, pm, offx, lane, o16, nsrc);
            roundtrip_half<VARIANT>(smem, strip + kPiecesPerChunk, Y, lm1, pm, offx, lane, o16 + kPiecesPerChunk, nsrc + kPiecesPerChunk);
            wave_lds_fence();
            {
                const uint4 *row = reinterpret_cast<const uint4 *>(strip + lane * kPiecesPerFrame);
                uint64_t s = 0;
                uint32_t peak = 0, bsum = 0, fail = 0;
#pragma unroll
                for (int i = 0; i < kPiecesPerFrame / 2; ++i) {
                    const uint4 v = row[i];
                    s += (uint64_t)(v.x + v.z);
                    peak = max(max(peak, v.y & 0x7FFFu), v.w & 0x7FFFu);
                    bsum += ((v.y >> 16) & 0x7FFFu) + ((v.w >> 16) & 0x7FFFu);
                    fail |= v.y | v.w;
                }
                uint32_t bm, fl;
                st_stream(reinterpret_cast<uint4 *>(stats + ((uint64_t)f * C + cme)), pack_stats160(s, peak, bsum, my_alaw, (fail >> 31) == 0u, bm, fl));
                if (open) {
                    h.sumsq_acc += s << 4; h.count += 1u; h.level_sum += bm; h.samples += (uint32_t)kFrame;
                    h.peak_hold = (uint16_t)max((uint32_t)h.peak_hold, peak);
                    h.level_max = (uint8_t)max((uint32_t)h.level_max, bm);
                    h.level_min = (uint8_t)min((uint32_t)h.level_min, bm);
                    h.n_silent += (fl & IGDSP_FLAG_SILENT) ? 1u : 0u;
                    h.n_clipped += (fl & IGDSP_FLAG_CLIPPED) ? 1u : 0u;
                }
            }
            wave_lds_fence();
        }
        if (n_seg == 1u) hold[cme] = h;
        else if (h.count != 0u) hold_merge(hold + cme, h);
    }
}

// ============================================================================
// Config #5, default form — k_roundtrip_lut64: the same channel-group-major walk as k_roundtrip_chunk64 with the
// compressor folded INTO the expansion LUT.  A G.711 code has 256 values per law, so decode -> re-encode is a function
// of (law, code): at kernel start every block evaluates the real expander and the real compressor (enc_uni, the
// arithmetic igdsp_encode runs) on each of the 2 x 128 magnitudes, once for +|x| and once for -|x|, and stores
//     entry = { (|x|/4)^2 ,  enc(-|x|) | enc(+|x|) << 8 | |x| << 16 }
// in the 32-replica conflict-free layout of fill_lut.  Per sample the kernel then does ONE ds_read_b64 (as the meter) and
// the re-encoded byte is picked by the code's sign bit with v_perm_b32 (3 perms + 2 VALU per 4 samples); the second LDS
// read per sample of the cell-table form (whose bank conflicts kept the LDS 80 % busy) is gone, and the
// frame peak comes from max(entry.y) >> 16 (the low bytes only break ties).  mu-law 0x7F ("-0") decodes to PCM 0 and
// re-encodes as enc(0) = 0xFF, exactly as two's-complement PCM between a real decoder and encoder would.
// ============================================================================
#ifndef IGDSP_RTL_WAVES
#define IGDSP_RTL_WAVES 12
#endif
constexpr int kRtlWaves = IGDSP_RTL_WAVES;

// Every one of the 256 entries is evaluated ONCE per block (two compressor evaluations each) into its replica 0 and copied to the
// other 31 replicas from there: with the few waves the block-owned round trip runs (6 x 64 threads) evaluating all 8 192 slots took
// 43 pairs of enc_uni per thread and launch.  Contains a barrier: call it from all threads; the caller's own barrier follows.
template <int VARIANT>
__device__ __forceinline__ void fill_recode_lut(uint2 *lut)
{
    const EncK ku = enc_consts<VARIANT>(false), ka = enc_consts<VARIANT>(true);
    for (uint32_t e = threadIdx.x; e < 256u; e += blockDim.x) {          // e = law<<7 | code7
        const bool alaw = (e & 0x80u) != 0u;
        const uint32_t ax = alaw ? alaw_abs(e) : ulaw_abs(e);
        const uint32_t m = ax >> 2;
        const uint32_t en = enc_uni<VARIANT>(-(int)ax, alaw ? ka : ku), ep = enc_uni<VARIANT>((int)ax, alaw ? ka : ku);
        lut[e << 5] = make_uint2(m * m, en | (ep << 8) | (ax << 16));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)kLutEntries; i += blockDim.x)
        if ((i & 31u) != 0u) lut[i] = lut[i & ~31u];
}

// Per-lane piece constants of a half, packed: five 5-bit frame indices (frame-in-half of piece j) in `fr5`, five 5-bit probe
// shifts in `pm5` (the probe byte a piece is responsible for sits at that bit of probe_fail's gathered word; 24 = none, that
// byte of the word is always zero).  Two registers instead of ten; one v_bfe_u32 (+ one shift) per piece to unpack.
__device__ __forceinline__ void pack_piece_consts(uint32_t lane, uint32_t &fr5, uint32_t &pm5)
{
    fr5 = 0; pm5 = 0;
#pragma unroll
    for (int j = 0; j < kLoadsPerChunk; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane, f = p / 10u, q = p - f * 10u;
        const uint32_t sh = q == 1u ? 0u : (q == 3u ? 8u : (q == 2u ? 16u : 24u));
        fr5 |= f << (5 * j);
        pm5 |= sh << (5 * j);
    }
}

// One half (32 frames) of a super-chunk: expand, meter, re-encode.  Same software pipeline as process_half (the LUT reads
// of unit u + 1 are in flight while unit u is folded); the eight re-encoded bytes of a unit are assembled right in its fold.
// All memory traffic goes through buffer instructions: `rin` describes the NEXT frame's super-chunk (refill), `rout` this
// frame's output super-chunk; `voff` = lane * 16, `hoff` = byte offset of the half inside the super-chunk.
__device__ __forceinline__ void recode_half(const uint2 *lut, uint2 *strip_half, uint4 (&d)[kLoadsPerChunk], const uint32_t am,
                                            const uint32_t fr5, const uint32_t pm5, const uint32_t off, const uint32_t lane,
                                            const uint32_t voff, const uint32_t hoff, __amdgpu_buffer_rsrc_t rin, __amdgpu_buffer_rsrc_t rout)
{
    uint2 e[2][8];
    uint32_t wa[2], wb[2];
    auto issue = [&](int u) {
        const int j = u >> 1, k = u & 1;
        wa[k] = (u & 1) ? d[j].z : d[j].x;
        wb[k] = (u & 1) ? d[j].w : d[j].y;
        const uint32_t frj = __builtin_amdgcn_ubfe(fr5, 5 * j, 5);
        const uint32_t lmj = (uint32_t)__builtin_amdgcn_sbfe(am, frj, 1) & 0x80808080u;
        const uint32_t ta = (wa[k] & 0x7F7F7F7Fu) | lmj, tb = (wb[k] & 0x7F7F7F7Fu) | lmj;
        e[k][0] = lut_at(lut, ta, off, 0x0C0C0400u); e[k][1] = lut_at(lut, ta, off, 0x0C0C0500u);
        e[k][2] = lut_at(lut, ta, off, 0x0C0C0600u); e[k][3] = lut_at(lut, ta, off, 0x0C0C0700u);
        e[k][4] = lut_at(lut, tb, off, 0x0C0C0400u); e[k][5] = lut_at(lut, tb, off, 0x0C0C0500u);
        e[k][6] = lut_at(lut, tb, off, 0x0C0C0600u); e[k][7] = lut_at(lut, tb, off, 0x0C0C0700u);
    };
    // four re-encoded bytes of word w: entries' byte 0 = enc(-|x|), byte 1 = enc(+|x|); a code is positive iff its bit 7 is set
    auto recode4 = [&](uint32_t w, const uint2 &e0, const uint2 &e1, const uint2 &e2, const uint2 &e3) {
        const uint32_t p01 = __builtin_amdgcn_perm(e1.y, e0.y, 0x05040100u);       // [e0.neg, e0.pos, e1.neg, e1.pos]
        const uint32_t p23 = __builtin_amdgcn_perm(e3.y, e2.y, 0x05040100u);
        const uint32_t sel = ((w >> 7) & 0x01010101u) | 0x06040200u;              // byte i picks pair i, +1 when positive
        return __builtin_amdgcn_perm(p23, p01, sel);
    };
    uint32_t sum = 0, peak = 0, bsum = 0;
    uint32_t o[4];
    issue(0);
#pragma unroll
    for (int u = 0; u < 2 * kLoadsPerChunk; ++u) {
        const int j = u >> 1, k = u & 1;
        if (u + 1 < 2 * kLoadsPerChunk) issue(u + 1);
        __builtin_amdgcn_sched_barrier(0);
        bsum = __builtin_amdgcn_sad_u8(wa[k], 0u, bsum);
        bsum = __builtin_amdgcn_sad_u8(wb[k], 0u, bsum);
        sum = sum + e[k][0].x + e[k][1].x; sum = sum + e[k][2].x + e[k][3].x;
        sum = sum + e[k][4].x + e[k][5].x; sum = sum + e[k][6].x + e[k][7].x;
        peak = max(max(peak, e[k][0].y), e[k][1].y); peak = max(max(peak, e[k][2].y), e[k][3].y);   // |x| << 16 dominates the compare
        peak = max(max(peak, e[k][4].y), e[k][5].y); peak = max(max(peak, e[k][6].y), e[k][7].y);
        o[2 * k] = recode4(wa[k], e[k][0], e[k][1], e[k][2], e[k][3]);
        o[2 * k + 1] = recode4(wb[k], e[k][4], e[k][5], e[k][6], e[k][7]);
        if (k == 1) {
            const uint32_t pmj = 0xFFu << __builtin_amdgcn_ubfe(pm5, 5 * j, 5);
            strip_half[j * 64 + lane] = make_uint2(sum, (peak >> 16) | (bsum << 16) | probe_fail(d[j], pmj));
            buf_st(rout, voff, hoff + (uint32_t)j * 1024u, make_uint4(o[0], o[1], o[2], o[3]));
            d[j] = buf_ld_stream(rin, voff, hoff + (uint32_t)j * 1024u);
            sum = 0; peak = 0; bsum = 0;
        }
    }
}

template <int VARIANT>
__global__ __launch_bounds__(kRtlWaves * 64) void k_roundtrip_lut64(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t F,
    uint8_t *__restrict__ out, igdsp_frame_stats *__restrict__ stats, igdsp_chan_hold *__restrict__ hold,
    const uint8_t *__restrict__ gate, uint32_t n_seg, uint32_t n_groups, uint32_t order)
{
    // Work item = (group of 64 consecutive channels, segment of the F frames), as in k_roundtrip_chunk64: n_seg == 1 ->
    // the wave owns hold[c] outright; n_seg > 1 -> windows merge by device-scope integer atomics (exact, order-free).
    // n_groups = C / 64 channel groups are handled here; channels beyond 64 * n_groups (C % 64) belong to the general kernel.
    __shared__ uint2 lds[kLutEntries + kRtlWaves * kStripEntries];
    fill_recode_lut<VARIANT>(lds);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint2 *strip = lds + kLutEntries + wave * kStripEntries;
    const uint32_t off = (lane & 31u) * 8u, voff = lane * 16u;
    uint32_t fr5, pm5;
    pack_piece_consts(lane, fr5, pm5);
    const uint32_t total_waves = gridDim.x * kRtlWaves;
    const uint64_t fbytes = (uint64_t)C * kFrame;                  // bytes between two frames of one channel group

    // order 0: the waves of a block take items a grid apart (neighbouring BLOCKS touch neighbouring groups); order 1 (default):
    // in the FIRST round the waves of a block take consecutive items (one block touches kRtlWaves neighbouring groups = 120 KiB
    // per frame: 0.4695 vs 0.4756 ms in an alternating in-process A/B, tools/rt_knobs.py); later rounds — the remainder when the
    // item count is not a multiple of the wave count — always go a grid apart, which spreads them evenly over the CUs (with
    // consecutive items there, a few blocks got all of the remainder: 0.5745 ms at four segments).
    const uint32_t n_items = n_groups * n_seg;
    const uint32_t near = blockIdx.x * (uint32_t)kRtlWaves + wave, apart = wave * gridDim.x + blockIdx.x;
    for (uint32_t round = 0;; ++round) {
        const uint64_t item64 = (uint64_t)round * total_waves + ((order && round == 0u) ? near : apart);
        if (item64 >= n_items) { if ((uint64_t)round * total_waves >= n_items) break; else continue; }
        const uint32_t item = (uint32_t)item64;
        const uint32_t seg = item / n_groups, cg = item - seg * n_groups;
        const uint32_t f_lo = (uint32_t)(((uint64_t)F * seg) / n_seg), f_hi = (uint32_t)(((uint64_t)F * (seg + 1u)) / n_seg);
        if (f_lo >= f_hi) continue;
        const uint32_t c0 = cg * kSuperFrames, cme = c0 + lane;
        const bool my_alaw = codec[cme] == IGDSP_PT_PCMA;
        const bool open = (gate == nullptr) || (gate[cme] != 0);
        // the window of this item, packed: {peak_hold | level_max << 16} (packed 16-bit max), level_min, {n_silent | n_clipped << 16}
        // (a segment never has 65 536 frames: launcher), level_sum, sumsq; count = frames of the segment if the gate is open
        uint64_t h_sumsq = 0;
        uint32_t h_pm = 0, h_min = 255u, h_sc = 0, h_lsum = 0;
        const uint64_t amask = __ballot(my_alaw);
        const uint32_t am_lo = (uint32_t)amask, am_hi = (uint32_t)(amask >> 32);
        const uint8_t *in0 = payload + (uint64_t)c0 * kFrame;     // wave-uniform bases: frame f of this group sits f * fbytes further
        uint8_t *out0 = out + (uint64_t)c0 * kFrame;

        uint4 X[kLoadsPerChunk], Y[kLoadsPerChunk];
        {
            const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in0 + (uint64_t)f_lo * fbytes);
#pragma unroll
            for (int j = 0; j < kLoadsPerChunk; ++j) X[j] = buf_ld_stream(r0, voff, (uint32_t)j * 1024u);
#pragma unroll
            for (int j = 0; j < kLoadsPerChunk; ++j) Y[j] = buf_ld_stream(r0, voff, (uint32_t)kChunkBytes + (uint32_t)j * 1024u);
        }
        for (uint32_t f = f_lo; f < f_hi; ++f) {
            const bool more = f + 1u < f_hi;                    // wave-uniform; the last frame re-reads itself (cache hit)
            const __amdgpu_buffer_rsrc_t rin = make_rsrc(in0 + (uint64_t)(more ? f + 1u : f) * fbytes);
            const __amdgpu_buffer_rsrc_t rout = make_rsrc(out0 + (uint64_t)f * fbytes);
            recode_half(lds, strip, X, am_lo, fr5, pm5, off, lane, voff, 0u, rin, rout);
            recode_half(lds, strip + kPiecesPerChunk, Y, am_hi, fr5, pm5, off, lane, voff, (uint32_t)kChunkBytes, rin, rout);
            wave_lds_fence();
            {
                const uint4 *row = reinterpret_cast<const uint4 *>(strip + lane * kPiecesPerFrame);
                uint64_t sq = 0;
                uint32_t peak = 0, bsum = 0, fail = 0;
#pragma unroll
                for (int i = 0; i < kPiecesPerFrame / 2; ++i) {
                    const uint4 v = row[i];
                    sq += (uint64_t)(v.x + v.z);
                    peak = max(max(peak, v.y & 0x7FFFu), v.w & 0x7FFFu);
                    bsum += ((v.y >> 16) & 0x7FFFu) + ((v.w >> 16) & 0x7FFFu);
                    fail |= v.y | v.w;
                }
                uint32_t bm, fl;
                const uint4 rec = pack_stats160(sq, peak, bsum, my_alaw, (fail >> 31) == 0u, bm, fl);
                buf_st(make_rsrc(stats + ((uint64_t)f * C + c0)), voff, 0u, rec);       // 64 records = 1 KiB, lane * 16
                h_sumsq += sq << 4; h_lsum += bm;
                h_pm = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2u16_t, h_pm), __builtin_bit_cast(v2u16_t, peak | (bm << 16))));
                h_min = min(h_min, bm);
                h_sc += ((fl & IGDSP_FLAG_SILENT) ? 1u : 0u) + ((fl & IGDSP_FLAG_CLIPPED) ? 0x10000u : 0u);
            }
            wave_lds_fence();
        }
        if (open) {
            igdsp_chan_hold h;
            const uint32_t cnt = f_hi - f_lo;
            h.sumsq_acc = h_sumsq; h.count = cnt; h.level_sum = h_lsum; h.samples = cnt * (uint32_t)kFrame;
            h.peak_hold = (uint16_t)(h_pm & 0xFFFFu); h.level_max = (uint8_t)(h_pm >> 16); h.level_min = (uint8_t)h_min;
            h.n_silent = h_sc & 0xFFFFu; h.n_clipped = h_sc >> 16;
            if (n_seg == 1u) {                                   // the wave owns hold[c]: plain read-modify-write
                igdsp_chan_hold g = hold[cme];
                g.sumsq_acc += h.sumsq_acc; g.count += h.count; g.level_sum += h.level_sum; g.samples += h.samples;
                g.peak_hold = max(g.peak_hold, h.peak_hold); g.level_max = max(g.level_max, h.level_max); g.level_min = min(g.level_min, h.level_min);
                g.n_silent += h.n_silent; g.n_clipped += h.n_clipped;
                hold[cme] = g;
            } else hold_merge(hold + cme, h);
        }
    }
}

// ============================================================================
// Config #5, block-owned form — k_roundtrip_blk64 (late round 3): k_roundtrip_lut64's item body under the work distribution of the
// block-owned window kernel (k_meter_rtp64<WIN = 2>, igdsp_k_packets.hip).  Block b owns the gpb = 4 (2, 1) consecutive channel
// groups b * gpb ... for the launch and hands their F * gpb items — one frame of one group: 10 KiB in, 10 KiB out, 1 KiB of
// records — to its waves one at a time from an LDS counter: a wave that falls behind draws fewer items, where the static walk
// gives every wave a fixed third of a group's frames.  The windows of the block's <= 256 channels live in LDS and move by LDS
// atomics (integer add / max / min: exact, any order — this kernel has no order-dependent state); at its end the block folds
// them into hold[c] itself (it owns the channels: plain read-modify-write, hold[c] fetched one item ahead as in the window
// kernel).  Blocks with an odd index walk the frames from the middle of the launch: at any moment half the chip writes the
// first half of the output and half the second, which is what an output spread over two memory classes (IGDSP_IO_BULK)
// needs, as the item orders of k_roundtrip_lut64 do.
// ============================================================================
constexpr int kRtBlkCh = 256;                             // channels a block can own (7 dwords of LDS each)
constexpr int kRtbWaves = 12;                             // strips for up to 12 waves; the launcher starts fewer (rtb_waves)
constexpr int kRtsbWaves = 10;                            // the same for k_roundtrip_strided<BLK> (its strips and rings are larger)

template <int VARIANT>
__global__ __launch_bounds__(kRtbWaves * 64) void k_roundtrip_blk64(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t F,
    uint8_t *__restrict__ out, igdsp_frame_stats *__restrict__ stats, igdsp_chan_hold *__restrict__ hold,
    const uint8_t *__restrict__ gate, uint32_t gpb, uint32_t gsh, uint32_t mid_start)
{
    __shared__ uint2 lds[kLutEntries + kRtbWaves * kStripEntries];
    // {sum of squares (2 dwords), byte-mean sum, peak-hold, level max, level min, silent | clipped << 16} x kRtBlkCh channels
    __shared__ uint32_t wst[7 * kRtBlkCh];
    __shared__ uint32_t q_next, q_ticket;
    fill_recode_lut<VARIANT>(lds);
    for (uint32_t i = threadIdx.x; i < 7u * (uint32_t)kRtBlkCh; i += blockDim.x) wst[i] = (i >= 5u * (uint32_t)kRtBlkCh && i < 6u * (uint32_t)kRtBlkCh) ? 255u : 0u;
    if (threadIdx.x == 0) { q_next = 0u; q_ticket = 0u; }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint2 *strip = lds + kLutEntries + wave * kStripEntries;
    const uint32_t off = (lane & 31u) * 8u, voff = lane * 16u;
    uint32_t fr5, pm5;
    pack_piece_consts(lane, fr5, pm5);
    const uint64_t fbytes = (uint64_t)C * kFrame;                  // bytes between two frames of one channel group
    const uint32_t b_first = blockIdx.x * gpb, b_items = F * gpb;
    const uint32_t f_shift = (mid_start != 0u && (blockIdx.x & 1u)) ? F / 2u : 0u;      // odd blocks start in the middle of the launch (spread outputs)
    auto grab = [&]() -> uint32_t {                                // block-local item number, 0xFFFFFFFF = none left
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&q_next, 1u);
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        return v < b_items ? v : 0xFFFFFFFFu;
    };
    auto frame_of = [&](uint32_t id) { const uint32_t f = (id >> gsh) + f_shift; return f >= F ? f - F : f; };
    auto group_of = [&](uint32_t id) { return b_first + (id & (gpb - 1u)); };
    // hold[c] of one group, fetched by the first gpb waves to start their last item (a global load at the block's end waits ~5 us)
    igdsp_chan_hold e_hold = igdsp_chan_hold{};
    uint32_t e_ticket = 0xFFFFFFFFu;
    bool e_open = false;
    auto end_prefetch = [&]() {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&q_ticket, 1u);
        e_ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        if (e_ticket < gpb) {
            const uint32_t c = (b_first + e_ticket) * (uint32_t)kSuperFrames + lane;
            e_open = gate == nullptr || gate[c] != 0;
            if (e_open) e_hold = hold[c];
        }
    };

    uint32_t id_cur = grab();
    if (id_cur != 0xFFFFFFFFu) {
        uint4 X[kLoadsPerChunk], Y[kLoadsPerChunk];
        uint32_t cur_pt = codec[group_of(id_cur) * (uint32_t)kSuperFrames + lane];
        {
            const __amdgpu_buffer_rsrc_t r0 = make_rsrc(payload + (uint64_t)group_of(id_cur) * kSuperFrames * kFrame + (uint64_t)frame_of(id_cur) * fbytes);
#pragma unroll
            for (int j = 0; j < kLoadsPerChunk; ++j) X[j] = buf_ld_stream(r0, voff, (uint32_t)j * 1024u);
#pragma unroll
            for (int j = 0; j < kLoadsPerChunk; ++j) Y[j] = buf_ld_stream(r0, voff, (uint32_t)kChunkBytes + (uint32_t)j * 1024u);
        }
        uint32_t id_next = grab();
        for (;;) {
            const bool more = id_next != 0xFFFFFFFFu;              // wave-uniform; the last item re-reads itself (cache hit)
            if (!more) end_prefetch();                            // (once: the loop ends with this item)
            const uint32_t id_load = more ? id_next : id_cur;
            const uint32_t cg = group_of(id_cur), f = frame_of(id_cur), c0 = cg * (uint32_t)kSuperFrames;
            const uint32_t cg_n = group_of(id_load), f_n = frame_of(id_load);
            const bool my_alaw = cur_pt == IGDSP_PT_PCMA;
            const uint64_t amask = __ballot(my_alaw);
            const uint32_t nxt_pt = codec[cg_n * (uint32_t)kSuperFrames + lane];
            const __amdgpu_buffer_rsrc_t rin = make_rsrc(payload + (uint64_t)cg_n * kSuperFrames * kFrame + (uint64_t)f_n * fbytes);
            const __amdgpu_buffer_rsrc_t rout = make_rsrc(out + (uint64_t)c0 * kFrame + (uint64_t)f * fbytes);
            recode_half(lds, strip, X, (uint32_t)amask, fr5, pm5, off, lane, voff, 0u, rin, rout);
            recode_half(lds, strip + kPiecesPerChunk, Y, (uint32_t)(amask >> 32), fr5, pm5, off, lane, voff, (uint32_t)kChunkBytes, rin, rout);
            const uint32_t id_after = more ? grab() : 0xFFFFFFFFu;  // its LDS round trip hides under the fold below
            wave_lds_fence();
            {
                const uint4 *row = reinterpret_cast<const uint4 *>(strip + lane * kPiecesPerFrame);
                uint64_t sq = 0;
                uint32_t peak = 0, bsum = 0, fail = 0;
#pragma unroll
                for (int i = 0; i < kPiecesPerFrame / 2; ++i) {
                    const uint4 v = row[i];
                    sq += (uint64_t)(v.x + v.z);
                    peak = max(max(peak, v.y & 0x7FFFu), v.w & 0x7FFFu);
                    bsum += ((v.y >> 16) & 0x7FFFu) + ((v.w >> 16) & 0x7FFFu);
                    fail |= v.y | v.w;
                }
                uint32_t bm, fl;
                const uint4 rec = pack_stats160(sq, peak, bsum, my_alaw, (fail >> 31) == 0u, bm, fl);
                buf_st(make_rsrc(stats + ((uint64_t)f * C + c0)), voff, 0u, rec);       // 64 records = 1 KiB, lane * 16
                const uint32_t cl = (id_cur & (gpb - 1u)) * (uint32_t)kSuperFrames + lane;
                __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(wst) + cl, (unsigned long long)(sq << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(wst + 2 * kRtBlkCh + cl, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(wst + 3 * kRtBlkCh + cl, peak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(wst + 4 * kRtBlkCh + cl, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_min(wst + 5 * kRtBlkCh + cl, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t sc = ((fl & IGDSP_FLAG_SILENT) ? 1u : 0u) + ((fl & IGDSP_FLAG_CLIPPED) ? 0x10000u : 0u);
                if (sc != 0u) __hip_atomic_fetch_add(wst + 6 * kRtBlkCh + cl, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            wave_lds_fence();
            if (!more) break;
            id_cur = id_next; id_next = id_after; cur_pt = nxt_pt;
        }
    }
    if (e_ticket == 0xFFFFFFFFu) end_prefetch();                   // a wave that never had an item
    __syncthreads();
    if (e_ticket < gpb && e_open) {
        const uint32_t t = e_ticket * (uint32_t)kSuperFrames + lane, c = b_first * (uint32_t)kSuperFrames + t;
        igdsp_chan_hold g = e_hold;
        const uint32_t sc = wst[6 * kRtBlkCh + t];
        g.sumsq_acc += reinterpret_cast<const unsigned long long *>(wst)[t]; g.count += F; g.level_sum += wst[2 * kRtBlkCh + t]; g.samples += F * (uint32_t)kFrame;
        g.peak_hold = (uint16_t)max((uint32_t)g.peak_hold, wst[3 * kRtBlkCh + t]);
        g.level_max = (uint8_t)max((uint32_t)g.level_max, wst[4 * kRtBlkCh + t]);
        g.level_min = (uint8_t)min((uint32_t)g.level_min, wst[5 * kRtBlkCh + t]);
        g.n_silent += sc & 0xFFFFu; g.n_clipped += sc >> 16;
        hold[c] = g;
    }
}

// ============================================================================
// Config #5 at the reference's other frame sizes — k_roundtrip_strided<Q, TAIL>: k_roundtrip_lut64's channel-group-major walk
// (hold window in registers, frame segments, atomic merge) over frames of n = 16 Q + 4 T bytes with k_meter_strided's piece
// geometry: Q + TAIL pieces per frame fetched at dword alignment through buffer instructions, the tail piece handing the
// frame's last dwords raw to the frame lane (stats) and storing their re-encoded bytes itself (output).
// ============================================================================
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// BLK (late round 3): the work distribution of k_roundtrip_blk64 — block b owns n_seg (= gpb) consecutive channel groups for the
// launch, hands their F * gpb items to its waves one at a time, keeps the windows in LDS (integer atomics) and folds them into
// hold[c] itself; odd blocks walk the frames from the middle of the launch.  n_seg carries gpb and order log2(gpb) then.
template <int Q, bool TAIL, int VARIANT, bool BLK = false>
__global__ __launch_bounds__((BLK ? (Q <= 1 ? 16 : kRtsbWaves) : kRtlWaves) * 64) void k_roundtrip_strided(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t F, uint32_t n,
    uint8_t *__restrict__ out, igdsp_frame_stats *__restrict__ stats, igdsp_chan_hold *__restrict__ hold,
    const uint8_t *__restrict__ gate, uint32_t n_seg, uint32_t n_groups, uint32_t order)
{
    constexpr int kW = BLK ? (Q <= 1 ? 16 : kRtsbWaves) : kRtlWaves;      // (16 .. 28-byte frames: small strips, 1.5 KiB items — as many waves as a block takes)
    static_assert(Q == 1 || Q >= 4, "the probe bytes 28 / 38 / 48 are taken from pieces 1 / 2 / 3");
    constexpr int QP = Q + (TAIL ? 1 : 0);
    constexpr int kStrip = kSuperFrames * QP;
    // RING (frames with a tail, n % 16 != 0): the re-encoded bytes of an item leave through a per-wave 2 KiB LDS ring as whole
    // 1 KiB runs at 1 KiB-aligned offsets of the item's output row (64 n bytes, a multiple of 128 for every n % 4 == 0 — so the
    // runs are whole cache lines).  Stored from the lanes' own registers a row of pieces is a contiguous ~1 KiB run too, but it
    // starts and ends anywhere: consecutive store instructions then share a line, written in two halves at different times
    // (round 2: 4 % more HBM traffic than the algorithm needs at n = 164, and 0.67 of peak where n = 160 runs at 0.75).
    constexpr bool RING = TAIL;
    __shared__ uint2 lds[kLutEntries + kW * kStrip + (RING ? kW * 256 : 0)];
    __shared__ uint32_t wst[BLK ? 7 * kRtBlkCh : 1];           // BLK: the windows of the block's channels, as in k_roundtrip_blk64
    __shared__ uint32_t q_next, q_ticket;
    fill_recode_lut<VARIANT>(lds);
    if (BLK) {
        for (uint32_t i = threadIdx.x; i < 7u * (uint32_t)kRtBlkCh; i += blockDim.x) wst[i] = (i >= 5u * (uint32_t)kRtBlkCh && i < 6u * (uint32_t)kRtBlkCh) ? 255u : 0u;
        if (threadIdx.x == 0) { q_next = 0u; q_ticket = 0u; }
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint2 *strip = lds + kLutEntries + wave * kStrip;
    uint32_t *ring = reinterpret_cast<uint32_t *>(lds + kLutEntries + kW * kStrip + (RING ? wave * 256u : 0u));   // 512 dwords
    const uint32_t off = (lane & 31u) * 8u;
    const uint32_t T = (n - 16u * Q) >> 2;
    constexpr int kPk = (QP + 1) / 2;
    uint32_t pk[kPk];
    uint32_t po[QP];                                             // byte offset of this lane's piece j inside a 64-channel frame row (registers:
#pragma unroll                                                   // recomputing f * n + ... per use cost two quarter-rate multiplies per piece)
    for (int j = 0; j < kPk; ++j) pk[j] = 0;
#pragma unroll
    for (int j = 0; j < QP; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane, f = p / (uint32_t)QP, q = p - f * (uint32_t)QP;
        const uint32_t sh = q == 1u ? 0u : (q == 3u ? 8u : (q == 2u ? 16u : 24u));
        const bool tp = TAIL && q == (uint32_t)Q;
        pk[j >> 1] |= (f | ((tp ? 24u : sh) << 8) | (tp ? 0x2000u : 0u)) << (16 * (j & 1));
        po[j] = f * n + (tp ? n - 16u : 16u * q);
    }
    auto fr_of = [&](int j) { return __builtin_amdgcn_ubfe(pk[j >> 1], 16 * (j & 1), 6); };
    auto ps_of = [&](int j) { return __builtin_amdgcn_ubfe(pk[j >> 1], 16 * (j & 1) + 8, 5); };
    auto tail_of = [&](int j) { return TAIL && ((pk[j >> 1] >> (16 * (j & 1) + 13)) & 1u) != 0u; };
    auto po_of = [&](int j) { return po[j]; };
    const uint32_t total_waves = gridDim.x * kW;
    const uint64_t fbytes = (uint64_t)C * n;                       // bytes between two frames of one channel group

    // one frame of one channel group: d holds its pieces, rin describes the item whose pieces refill d, acc takes the frame's figures
    auto frame_body = [&](const uint32_t c0, const uint32_t f, const __amdgpu_buffer_rsrc_t rin, const bool my_alaw, const uint32_t am_lo,
                          const uint32_t am_hi, uint4 (&d)[QP], auto &&acc) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rout = make_rsrc_ranged(out + (uint64_t)c0 * n + (uint64_t)f * fbytes, (uint32_t)kSuperFrames * n);   // this group's 64 frames of the row
        uint32_t stored = 0, pend_off = 0xFFFFFFFFu;         // RING: bytes of this item's output row already stored (a multiple of 1 KiB); the run read last row
        uint4 pend = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < kPk; ++j) asm volatile("" : "+v"(pk[j]));
        {
            uint2 e[2][8];
            uint32_t wa[2], wb[2];
            auto issue = [&](int u) {
                const int j = u >> 1, k = u & 1;
                wa[k] = (u & 1) ? d[j].z : d[j].x;
                wb[k] = (u & 1) ? d[j].w : d[j].y;
                const uint32_t frj = fr_of(j);
                const uint32_t bit = frj < 32u ? (uint32_t)__builtin_amdgcn_sbfe(am_lo, frj, 1) : (uint32_t)__builtin_amdgcn_sbfe(am_hi, frj - 32u, 1);
                const uint32_t lmj = bit & 0x80808080u;
                const uint32_t ta = (wa[k] & 0x7F7F7F7Fu) | lmj, tb = (wb[k] & 0x7F7F7F7Fu) | lmj;
                e[k][0] = lut_at(lds, ta, off, 0x0C0C0400u); e[k][1] = lut_at(lds, ta, off, 0x0C0C0500u);
                e[k][2] = lut_at(lds, ta, off, 0x0C0C0600u); e[k][3] = lut_at(lds, ta, off, 0x0C0C0700u);
                e[k][4] = lut_at(lds, tb, off, 0x0C0C0400u); e[k][5] = lut_at(lds, tb, off, 0x0C0C0500u);
                e[k][6] = lut_at(lds, tb, off, 0x0C0C0600u); e[k][7] = lut_at(lds, tb, off, 0x0C0C0700u);
            };
            auto recode4 = [&](uint32_t w, const uint2 &e0, const uint2 &e1, const uint2 &e2, const uint2 &e3) {
                const uint32_t p01 = __builtin_amdgcn_perm(e1.y, e0.y, 0x05040100u);
                const uint32_t p23 = __builtin_amdgcn_perm(e3.y, e2.y, 0x05040100u);
                const uint32_t sel = ((w >> 7) & 0x01010101u) | 0x06040200u;
                return __builtin_amdgcn_perm(p23, p01, sel);
            };
            uint32_t sum = 0, peak = 0, bsum = 0;
            uint32_t o[4];
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
                peak = max(max(peak, e[k][0].y), e[k][1].y); peak = max(max(peak, e[k][2].y), e[k][3].y);
                peak = max(max(peak, e[k][4].y), e[k][5].y); peak = max(max(peak, e[k][6].y), e[k][7].y);
                o[2 * k] = recode4(wa[k], e[k][0], e[k][1], e[k][2], e[k][3]);
                o[2 * k + 1] = recode4(wb[k], e[k][4], e[k][5], e[k][6], e[k][7]);
                if (k == 1) {
                    uint2 ent = make_uint2(sum, (peak >> 16) | (bsum << 16) | probe_fail(d[j], 0xFFu << ps_of(j)));
                    const uint32_t pj = po_of(j);
                    // no branch on the per-lane tail flag: every lane issues both stores, the one that does not apply at an offset past
                    // the frame row's buffer range (raw buffer stores out of range are dropped by the hardware)
                    const bool tl = tail_of(j);
                    if (tl) ent = make_uint2(d[j].z, d[j].w);               // the frame's last two dwords, raw, for the frame lane
                    if (!RING) buf_st(rout, pj, 0u, make_uint4(o[0], o[1], o[2], o[3]));
                    else {
                        // this piece's bytes at their offset S of the output row (a payload piece: 16 bytes at its own offset; the
                        // tail piece: the frame's last 4 T bytes, right behind piece Q - 1), parked in the ring dword by dword
                        // (S is only dword aligned); then every whole KiB the row has reached goes out, 16 bytes per lane
                        uint32_t S = tl ? pj + 16u - 4u * T : pj;
                        asm volatile("" : "+v"(S));                  // derived per use: hoisted out of the frame loop the ring addresses of all rows cost 50 registers (and spilled)
                        if (!tl) { ring[(S >> 2) & 511u] = o[0]; ring[((S >> 2) + 1u) & 511u] = o[1]; ring[((S >> 2) + 2u) & 511u] = o[2]; ring[((S >> 2) + 3u) & 511u] = o[3]; }
                        else if (T == 2u) { ring[(S >> 2) & 511u] = o[2]; ring[((S >> 2) + 1u) & 511u] = o[3]; }
                        else ring[(S >> 2) & 511u] = o[3];
                        const uint32_t reach = (uint32_t)__builtin_amdgcn_readlane((int)(S + (tl ? 4u * T : 16u)), 63);   // end of the row's run
                        // LDS operations of one wave execute in order, so the read below sees the dwords just parked and no wait is
                        // needed between them — only the compiler must keep the order.  The run read in row j is stored in row j + 1
                        // (pend): waiting for it at once would wait for every LUT read of the next unit issued before it.
                        asm volatile("" ::: "memory");
                        if (pend_off != 0xFFFFFFFFu) buf_st(rout, pend_off + 16u * lane, 0u, pend);
                        pend_off = 0xFFFFFFFFu;
                        if (reach - stored >= 1024u) {               // wave-uniform; a row adds at most 1 KiB, so at most one run is due
                            pend = *reinterpret_cast<const uint4 *>(ring + (((stored >> 2) + 4u * lane) & 511u));
                            pend_off = stored;
                            stored += 1024u;
                        }
                        if (j == QP - 1) {                           // the item's last row: the pending run, then what is left (64 n % 1024 bytes, whole lines)
                            if (pend_off != 0xFFFFFFFFu) buf_st(rout, pend_off + 16u * lane, 0u, pend);
                            pend_off = 0xFFFFFFFFu;
                            const uint32_t left = (uint32_t)kSuperFrames * n - stored;
                            const uint4 v = *reinterpret_cast<const uint4 *>(ring + (((stored >> 2) + 4u * lane) & 511u));
                            buf_st(rout, 16u * lane < left ? stored + 16u * lane : 0x80000000u, 0u, v);
                        }
                        asm volatile("" ::: "memory");
                    }
                    strip[j * 64 + lane] = ent;
                    d[j] = buf_ld_pieces(rin, pj, 0u);
                    sum = 0; peak = 0; bsum = 0;
                }
            }
        }
        wave_lds_fence();
        {
            const uint2 *row = strip + lane * QP;
            uint64_t sq = 0;
            uint32_t peak = 0, bsum = 0, fail = 0, part = 0;
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                const uint2 v = row[i];
                part += v.x;
                if ((i & 3) == 3 || i == Q - 1) { sq += part; part = 0; }
                peak = max(peak, v.y & 0x7FFFu);
                bsum += (v.y >> 16) & 0x7FFFu;
                fail |= v.y;
            }
            if (TAIL) {
                const uint2 tv = row[Q];
                const uint32_t lm = my_alaw ? 0x80808080u : 0u;
                const uint32_t tws[2] = {T == 2u ? tv.x : tv.y, tv.y};
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if ((uint32_t)t < T) {
                        const uint32_t w = tws[t], tt = (w & 0x7F7F7F7Fu) | lm;
                        const uint2 e0 = lut_at(lds, tt, off, 0x0C0C0400u), e1 = lut_at(lds, tt, off, 0x0C0C0500u);
                        const uint2 e2 = lut_at(lds, tt, off, 0x0C0C0600u), e3 = lut_at(lds, tt, off, 0x0C0C0700u);
                        sq += (uint64_t)(e0.x + e1.x + e2.x + e3.x);
                        peak = max(peak, max(max(e0.y, e1.y), max(e2.y, e3.y)) >> 16);
                        bsum = __builtin_amdgcn_sad_u8(w, 0u, bsum);
                    }
            }
            uint32_t bm, fl;
            const uint4 rec = pack_stats(sq << 4, peak, bsum, n, my_alaw, (Q >= 4) && (fail >> 31) == 0u, bm, fl);
            buf_st(make_rsrc(stats + ((uint64_t)f * C + c0)), lane * 16u, 0u, rec);
            acc(sq << 4, peak, bm, fl);
        }
        wave_lds_fence();
    };
    if (BLK) {
        const uint32_t gpb = n_seg, gsh = order & 3u, b_first = blockIdx.x * gpb, b_items = F * gpb;
        const uint32_t f_shift = ((order & 4u) != 0u && (blockIdx.x & 1u)) ? F / 2u : 0u;      // (order bit 2: the output is spread over two classes)
        auto grab = [&]() -> uint32_t {
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&q_next, 1u);
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            return v < b_items ? v : 0xFFFFFFFFu;
        };
        auto frame_of = [&](uint32_t id) { const uint32_t f = (id >> gsh) + f_shift; return f >= F ? f - F : f; };
        auto group_of = [&](uint32_t id) { return b_first + (id & (gpb - 1u)); };
        auto in_of = [&](uint32_t id) { return make_rsrc(payload + (uint64_t)group_of(id) * kSuperFrames * n + (uint64_t)frame_of(id) * fbytes); };
        igdsp_chan_hold e_hold = igdsp_chan_hold{};
        uint32_t e_ticket = 0xFFFFFFFFu;
        bool e_open = false;
        auto end_prefetch = [&]() {                               // hold[c] of one group, fetched a whole item before the block's end
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&q_ticket, 1u);
            e_ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            if (e_ticket < gpb) {
                const uint32_t c = (b_first + e_ticket) * (uint32_t)kSuperFrames + lane;
                e_open = gate == nullptr || gate[c] != 0;
                if (e_open) e_hold = hold[c];
            }
        };
        uint32_t id_cur = grab();
        if (id_cur != 0xFFFFFFFFu) {
            uint4 d[QP];
            uint32_t cur_pt = codec[group_of(id_cur) * (uint32_t)kSuperFrames + lane];
            {
                const __amdgpu_buffer_rsrc_t r0 = in_of(id_cur);
#pragma unroll
                for (int j = 0; j < QP; ++j) d[j] = buf_ld_pieces(r0, po_of(j), 0u);
            }
            uint32_t id_next = grab();
            for (;;) {
                const bool more = id_next != 0xFFFFFFFFu;          // the last item re-reads itself (cache hit)
                if (!more) end_prefetch();
                const uint32_t id_load = more ? id_next : id_cur;
                const bool my_alaw = cur_pt == IGDSP_PT_PCMA;
                const uint64_t amask = __ballot(my_alaw);
                const uint32_t nxt_pt = codec[group_of(id_load) * (uint32_t)kSuperFrames + lane];
                const uint32_t id_after = more ? grab() : 0xFFFFFFFFu;
                const uint32_t cl = (id_cur & (gpb - 1u)) * (uint32_t)kSuperFrames + lane;
                frame_body(group_of(id_cur) * (uint32_t)kSuperFrames, frame_of(id_cur), in_of(id_load), my_alaw, (uint32_t)amask, (uint32_t)(amask >> 32), d,
                           [&](uint64_t s16, uint32_t peak, uint32_t bm, uint32_t fl) {
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(wst) + cl, (unsigned long long)s16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(wst + 2 * kRtBlkCh + cl, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(wst + 3 * kRtBlkCh + cl, peak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(wst + 4 * kRtBlkCh + cl, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_min(wst + 5 * kRtBlkCh + cl, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t sc = ((fl & IGDSP_FLAG_SILENT) ? 1u : 0u) + ((fl & IGDSP_FLAG_CLIPPED) ? 0x10000u : 0u);
                    if (sc != 0u) __hip_atomic_fetch_add(wst + 6 * kRtBlkCh + cl, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                });
                if (!more) break;
                id_cur = id_next; id_next = id_after; cur_pt = nxt_pt;
            }
        }
        if (e_ticket == 0xFFFFFFFFu) end_prefetch();               // a wave that never had an item
        __syncthreads();
        if (e_ticket < gpb && e_open) {
            const uint32_t t = e_ticket * (uint32_t)kSuperFrames + lane, c = b_first * (uint32_t)kSuperFrames + t;
            igdsp_chan_hold g = e_hold;
            const uint32_t sc = wst[6 * kRtBlkCh + t];
            g.sumsq_acc += reinterpret_cast<const unsigned long long *>(wst)[t]; g.count += F; g.level_sum += wst[2 * kRtBlkCh + t]; g.samples += F * n;
            g.peak_hold = (uint16_t)max((uint32_t)g.peak_hold, wst[3 * kRtBlkCh + t]);
            g.level_max = (uint8_t)max((uint32_t)g.level_max, wst[4 * kRtBlkCh + t]);
            g.level_min = (uint8_t)min((uint32_t)g.level_min, wst[5 * kRtBlkCh + t]);
            g.n_silent += sc & 0xFFFFu; g.n_clipped += sc >> 16;
            hold[c] = g;
        }
        return;
    }
    // item order as in k_roundtrip_lut64: consecutive groups per block in the first round when the output is spread over two classes
    const uint32_t n_items = n_groups * n_seg;
    const uint32_t near = blockIdx.x * (uint32_t)kW + wave, apart = wave * gridDim.x + blockIdx.x;
    for (uint32_t round = 0;; ++round) {
        const uint64_t item64 = (uint64_t)round * total_waves + ((order && round == 0u) ? near : apart);
        if (item64 >= n_items) { if ((uint64_t)round * total_waves >= n_items) break; else continue; }
        const uint32_t item = (uint32_t)item64;
        const uint32_t seg = item / n_groups, cg = item - seg * n_groups;
        const uint32_t f_lo = (uint32_t)(((uint64_t)F * seg) / n_seg), f_hi = (uint32_t)(((uint64_t)F * (seg + 1u)) / n_seg);
        if (f_lo >= f_hi) continue;
        const uint32_t c0 = cg * kSuperFrames, cme = c0 + lane;
        const bool my_alaw = codec[cme] == IGDSP_PT_PCMA;
        const bool open = (gate == nullptr) || (gate[cme] != 0);
        uint64_t h_sumsq = 0;
        uint32_t h_pm = 0, h_min = 255u, h_sc = 0, h_lsum = 0;
        const uint64_t amask = __ballot(my_alaw);
        const uint32_t am_lo = (uint32_t)amask, am_hi = (uint32_t)(amask >> 32);
        const uint8_t *in0 = payload + (uint64_t)c0 * n;

        uint4 d[QP];
        {
            const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in0 + (uint64_t)f_lo * fbytes);
#pragma unroll
            for (int j = 0; j < QP; ++j) d[j] = buf_ld_pieces(r0, po_of(j), 0u);
        }
        for (uint32_t f = f_lo; f < f_hi; ++f) {
            const bool more = f + 1u < f_hi;
            const __amdgpu_buffer_rsrc_t rin = make_rsrc(in0 + (uint64_t)(more ? f + 1u : f) * fbytes);
            frame_body(c0, f, rin, my_alaw, am_lo, am_hi, d, [&](uint64_t s16, uint32_t peak, uint32_t bm, uint32_t fl) {
                h_sumsq += s16; h_lsum += bm;
                h_pm = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2u16_t, h_pm), __builtin_bit_cast(v2u16_t, peak | (bm << 16))));
                h_min = min(h_min, bm);
                h_sc += ((fl & IGDSP_FLAG_SILENT) ? 1u : 0u) + ((fl & IGDSP_FLAG_CLIPPED) ? 0x10000u : 0u);
            });
        }
        if (open) {
            igdsp_chan_hold h;
            const uint32_t cnt = f_hi - f_lo;
            h.sumsq_acc = h_sumsq; h.count = cnt; h.level_sum = h_lsum; h.samples = cnt * n;
            h.peak_hold = (uint16_t)(h_pm & 0xFFFFu); h.level_max = (uint8_t)(h_pm >> 16); h.level_min = (uint8_t)h_min;
            h.n_silent = h_sc & 0xFFFFu; h.n_clipped = h_sc >> 16;
            if (n_seg == 1u) {
                igdsp_chan_hold g = hold[cme];
                g.sumsq_acc += h.sumsq_acc; g.count += h.count; g.level_sum += h.level_sum; g.samples += h.samples;
                g.peak_hold = max(g.peak_hold, h.peak_hold); g.level_max = max(g.level_max, h.level_max); g.level_min = min(g.level_min, h.level_min);
                g.n_silent += h.n_silent; g.n_clipped += h.n_clipped;
                hold[cme] = g;
            } else hold_merge(hold + cme, h);
        }
    }
}

// ============================================================================
// Config #5 for every other shape — k_roundtrip_general: one wavefront per CHANNEL walks that channel's F frames
// (any n in 1..256, any C, unaligned buffers), lane l owning bytes [4l, 4l + 4) of a frame as in
// k_meter_wave_per_frame.  Decode through the signed 256-entry LUT, stats by wave reduction, re-encode with the full
// compressor arithmetic (enc_uni) on the decoded PCM, window aggregate in registers, hold[c] written once.  It serves
// BASELINE config #1's 4 channels, 164- or 24-byte frames, and the C % 64 channels the fused kernel leaves over
// (channels [c_first, c_first + c_count) of a [F][C][n] batch).
// ============================================================================
template <int VARIANT>
__global__ __launch_bounds__(256) void k_roundtrip_general(
    const uint8_t *__restrict__ payload, const uint8_t *__restrict__ codec, uint32_t C, uint32_t F, uint32_t n,
    uint32_t c_first, uint32_t c_count, uint8_t *__restrict__ out, igdsp_frame_stats *__restrict__ stats,
    igdsp_chan_hold *__restrict__ hold, const uint8_t *__restrict__ gate)
{
    __shared__ int16_t lut[2][256];
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) {
        const uint32_t code = i & 255u;
        const int ax = (int)((i >> 8) ? alaw_abs(code) : ulaw_abs(code));
        lut[i >> 8][code] = (int16_t)((code & 0x80u) ? ax : -ax);
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool dword_ok = ((n & 3u) == 0u) && (((reinterpret_cast<uintptr_t>(payload) | reinterpret_cast<uintptr_t>(out)) & 3u) == 0u);
    const uint32_t b0 = lane * 4u;
    const uint32_t nvalid = (n > b0) ? min(n - b0, 4u) : 0u;
    for (uint32_t ci = blockIdx.x * 4u + wave; ci < c_count; ci += gridDim.x * 4u) {
        const uint32_t c = c_first + ci;
        const bool alaw = codec[c] == IGDSP_PT_PCMA;
        const bool open = (gate == nullptr) || (gate[c] != 0);
        const EncK ek = enc_consts<VARIANT>(alaw);
        igdsp_chan_hold h = hold[c];                              // wave-uniform copy; lane 0 writes it back
        auto load_frame = [&](uint32_t f) -> uint32_t {
            const uint8_t *base = payload + ((uint64_t)f * C + c) * n;
            uint32_t w = 0;
            if (dword_ok) { if (b0 < n) w = *reinterpret_cast<const uint32_t *>(base + b0); }
            else {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) if (k < nvalid) w |= (uint32_t)base[b0 + k] << (8u * k);
            }
            return w;
        };
        uint32_t w = load_frame(0);
        for (uint32_t f = 0; f < F; ++f) {
            const uint32_t wn = load_frame(min(f + 1u, F - 1u));  // next frame in flight while this one is folded
            uint32_t sum = 0, peak = 0, bsum = 0, o = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t b = (w >> (8u * k)) & 255u;
                const int v = (k < nvalid) ? (int)lut[alaw][b] : 0;
                const uint32_t ax = (uint32_t)(v < 0 ? -v : v);
                sum += (ax >> 2) * (ax >> 2);
                peak = max(peak, ax);
                bsum += (k < nvalid) ? b : 0u;
                o |= enc_uni<VARIANT>(v, ek) << (8u * k);        // the compressor on the decoded PCM value
            }
            if (b0 < n) {
                uint8_t *ob = out + ((uint64_t)f * C + c) * n + b0;
                if (dword_ok) *reinterpret_cast<uint32_t *>(ob) = o;
                else {
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) if (k < nvalid) ob[k] = (uint8_t)(o >> (8u * k));
                }
            }
            const uint32_t w7 = (uint32_t)__builtin_amdgcn_readlane((int)w, 7), w9 = (uint32_t)__builtin_amdgcn_readlane((int)w, 9),
                           w12 = (uint32_t)__builtin_amdgcn_readlane((int)w, 12);
            const bool probe = (n > 48u) && ((w7 & 255u) == 0xD5u) && (((w9 >> 16) & 255u) == 0xD5u) && ((w12 & 255u) == 0xD5u);
            const uint32_t r_lo = wave_reduce_dpp(sum & 0xFFFFu, OpAdd()), r_hi = wave_reduce_dpp(sum >> 16, OpAdd());
            const uint64_t s64 = (((uint64_t)r_hi << 16) + r_lo) << 4;
            peak = wave_reduce_dpp(peak, OpMax());
            bsum = wave_reduce_dpp(bsum, OpAdd());
            const igdsp_frame_stats st = make_stats(s64, peak, bsum, n, alaw, probe);
            if (lane == 0) stats[(uint64_t)f * C + c] = st;
            if (open) {
                h.sumsq_acc += s64; h.count += 1u; h.level_sum += st.byte_mean; h.samples += n;
                h.peak_hold = (uint16_t)max((uint32_t)h.peak_hold, peak);
                h.level_max = (uint8_t)max((uint32_t)h.level_max, (uint32_t)st.byte_mean);
                h.level_min = (uint8_t)min((uint32_t)h.level_min, (uint32_t)st.byte_mean);
                h.n_silent += (st.flags & IGDSP_FLAG_SILENT) ? 1u : 0u;
                h.n_clipped += (st.flags & IGDSP_FLAG_CLIPPED) ? 1u : 0u;
            }
            w = wn;
        }
        if (lane == 0) hold[c] = h;
    }
}

// a6 — fold stats[f][c] into hold[c]; one thread per channel, coalesced over c.
__global__ __launch_bounds__(256) void k_hold_update(const igdsp_frame_stats *__restrict__ stats,
                                                     const uint16_t *__restrict__ len, uint32_t C, uint32_t F,
                                                     uint32_t n, igdsp_chan_hold *__restrict__ hold,
                                                     const uint8_t *__restrict__ gate)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (gate != nullptr && gate[c] == 0) return;
    igdsp_chan_hold h = hold[c];
    for (uint32_t f = 0; f < F; ++f) {
        const igdsp_frame_stats s = stats[(uint64_t)f * C + c];
        if (s.flags & IGDSP_FLAG_EMPTY) continue;
        h.sumsq_acc += s.sumsq; h.count += 1u; h.level_sum += s.byte_mean;
        h.samples += len ? min((uint32_t)len[(uint64_t)f * C + c], n) : n;
        h.peak_hold = (uint16_t)max((uint32_t)h.peak_hold, (uint32_t)s.peak);
        h.level_max = (uint8_t)max((uint32_t)h.level_max, (uint32_t)s.byte_mean);
        h.level_min = (uint8_t)min((uint32_t)h.level_min, (uint32_t)s.byte_mean);
        h.n_silent += (s.flags & IGDSP_FLAG_SILENT) ? 1u : 0u;
        h.n_clipped += (s.flags & IGDSP_FLAG_CLIPPED) ? 1u : 0u;
    }
    hold[c] = h;
}

__device__ __forceinline__ bool frame_gate(uint32_t mode, uint32_t ed)
{
    const uint32_t squ = (ed >> 28) & 1u, ptt = ed >> 29;
    return mode == IGDSP_GATE_ALWAYS || (mode == IGDSP_GATE_SQU && squ != 0u) || (mode == IGDSP_GATE_PTT && ptt != 0u) ||
           (mode == IGDSP_GATE_SQU_OR_PTT && (squ | ptt) != 0u);
}

// a6 on the drop-in path: fold the records of one flush into the per-channel state.  The flush hands every channel its staged
// frames in ARRIVAL order — seq[first .. first + count) = {record id (group B: | 0x80000000), ED-137 word of the call when the
// frame was staged}; one thread per channel folds them: keeplogAudioLevel per frame (Functions.cpp:2126-2145) under the frame
// gate (PTT / SQU of the word, Functions.cpp:1136, 1160), the consecutive-silence run (adapter->rtpFalse,
// TransportAdapter.cpp:657-673), and the channel's newest record for igdsp_poll.
__global__ __launch_bounds__(256) void k_flush_fold(const igdsp_frame_stats *__restrict__ stA, const igdsp_frame_stats *__restrict__ stB,
                                                    const uint16_t *__restrict__ lenB, const uint2 *__restrict__ seq, const uint2 *__restrict__ runs,
                                                    uint32_t n_channels, uint32_t gate_mode, uint32_t alarm, igdsp_chan_hold *__restrict__ hold,
                                                    igdsp_chan_probe *__restrict__ probe, igdsp_frame_stats *__restrict__ last)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_channels) return;
    const uint2 run = runs[c];
    if (run.y == 0u) return;
    igdsp_chan_hold h = hold[c];
    igdsp_chan_probe p = probe[c];
    igdsp_frame_stats newest = last[c];
    for (uint32_t i = run.x; i < run.x + run.y; ++i) {
        const uint2 e = seq[i];
        const bool b = (e.x >> 31) != 0u;
        const uint32_t id = e.x & 0x7FFFFFFFu;
        const igdsp_frame_stats s = b ? stB[id] : stA[id];
        if (s.flags & IGDSP_FLAG_EMPTY) continue;
        const uint32_t l = b ? min((uint32_t)lenB[id], (uint32_t)IGDSP_MAX_PAYLOAD) : (uint32_t)IGDSP_SAMPLES_PER_FRAME;
        newest = s;
        if (l > 48u) {
            if (s.flags & IGDSP_FLAG_PROBE_D5) { p.run += 1u; p.alarms += (p.run == alarm) ? 1u : 0u; }
            else p.run = 0u;
        }
        if (!frame_gate(gate_mode, e.y)) continue;
        h.sumsq_acc += s.sumsq; h.count += 1u; h.level_sum += s.byte_mean; h.samples += l;
        h.peak_hold = (uint16_t)max((uint32_t)h.peak_hold, (uint32_t)s.peak);
        h.level_max = (uint8_t)max((uint32_t)h.level_max, (uint32_t)s.byte_mean);
        h.level_min = (uint8_t)min((uint32_t)h.level_min, (uint32_t)s.byte_mean);
        h.n_silent += (s.flags & IGDSP_FLAG_SILENT) ? 1u : 0u;
        h.n_clipped += (s.flags & IGDSP_FLAG_CLIPPED) ? 1u : 0u;
    }
    hold[c] = h;
    probe[c] = p;
    last[c] = newest;
}

__global__ __launch_bounds__(256) void k_hold_reset(igdsp_chan_hold *__restrict__ hold, uint32_t C,
                                                    const uint8_t *__restrict__ mask)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (mask != nullptr && mask[c] == 0) return;
    igdsp_chan_hold h;
    h.sumsq_acc = 0; h.count = 0; h.level_sum = 0; h.samples = 0; h.peak_hold = 0;
    h.level_max = 0; h.level_min = 255; h.n_silent = 0; h.n_clipped = 0;
    hold[c] = h;
}

// ED-137 gated window over records (igdsp_window_update): one thread per channel walks its F frames in order — the generalisation
// of k_hold_update to per-FRAME gates (ED-137 word of each frame's own packet: PTT type bits 31-29, SQU bit 28, masks
// Functions.cpp:1136, 1160) plus the consecutive-silence run (adapter->rtpFalse, TransportAdapter.cpp:657-673).
__global__ __launch_bounds__(256) void k_window_update(const igdsp_frame_stats *__restrict__ stats, const igdsp_rtp_info *__restrict__ info,
                                                       const uint16_t *__restrict__ len, uint32_t C, uint32_t F, uint32_t n,
                                                       uint32_t gate_mode, uint32_t alarm, igdsp_chan_hold *__restrict__ hold,
                                                       const uint8_t *__restrict__ gate, igdsp_chan_probe *__restrict__ probe)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const bool open = gate == nullptr || gate[c] != 0;
    igdsp_chan_hold h = hold[c];
    igdsp_chan_probe p = probe ? probe[c] : igdsp_chan_probe{0u, 0u};
    for (uint32_t f = 0; f < F; ++f) {
        const uint64_t fi = (uint64_t)f * C + c;
        const igdsp_frame_stats s = stats[fi];
        if (s.flags & IGDSP_FLAG_EMPTY) continue;
        uint32_t ed = 0, l = n;
        if (info != nullptr) { const igdsp_rtp_info r = info[fi]; ed = r.ed137; l = r.payload_len; }
        if (len != nullptr) l = len[fi];
        l = min(l, n);
        if (l > 48u) {
            if (s.flags & IGDSP_FLAG_PROBE_D5) { p.run += 1u; p.alarms += (p.run == alarm) ? 1u : 0u; }
            else p.run = 0u;
        }
        if (!open || !frame_gate(gate_mode, ed)) continue;
        h.sumsq_acc += s.sumsq; h.count += 1u; h.level_sum += s.byte_mean; h.samples += l;
        h.peak_hold = (uint16_t)max((uint32_t)h.peak_hold, (uint32_t)s.peak);
        h.level_max = (uint8_t)max((uint32_t)h.level_max, (uint32_t)s.byte_mean);
        h.level_min = (uint8_t)min((uint32_t)h.level_min, (uint32_t)s.byte_mean);
        h.n_silent += (s.flags & IGDSP_FLAG_SILENT) ? 1u : 0u;
        h.n_clipped += (s.flags & IGDSP_FLAG_CLIPPED) ? 1u : 0u;
    }
    hold[c] = h;
    if (probe) probe[c] = p;
}

// Fold the per-unit summaries of a fused window launch (k_meter_rtp64<WIN>) into hold[c] / probe[c], segments in frame order:
// work[seg][0][c] = {sumsq lo, sumsq hi, frames, byte-mean sum}, [1] = {samples, peak_hold | level_max << 16, level_min, n_silent |
// n_clipped << 16}, [2] = {probe frames before the segment's first reset (all of them if it had none), run at its end, alarms after
// its first reset, had a reset}.  The leading probe frames of a segment continue the incoming run and raise an alarm if they carry
// it across the alarm length.  gate[c] == 0 (the caller's per-channel window state) leaves hold[c] alone.
__global__ __launch_bounds__(256) void k_window_finish(const uint4 *__restrict__ work, uint32_t C, uint32_t n_seg, uint32_t alarm,
                                                       igdsp_chan_hold *__restrict__ hold, const uint8_t *__restrict__ gate,
                                                       igdsp_chan_probe *__restrict__ probe)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const bool open = gate == nullptr || gate[c] != 0;
    igdsp_chan_hold h = hold[c];
    igdsp_chan_probe p = probe ? probe[c] : igdsp_chan_probe{0u, 0u};
    for (uint32_t sg = 0; sg < n_seg; ++sg) {
        const uint4 *wk = work + ((uint64_t)sg * 3u * C + c);
        const uint4 a = wk[0], b = wk[C], r = wk[2u * (uint64_t)C];
        if (open && a.z != 0u) {
            h.sumsq_acc += ((uint64_t)a.y << 32) | a.x; h.count += a.z; h.level_sum += a.w; h.samples += b.x;
            h.peak_hold = (uint16_t)max((uint32_t)h.peak_hold, b.y & 0xFFFFu);
            h.level_max = (uint8_t)max((uint32_t)h.level_max, b.y >> 16);
            h.level_min = (uint8_t)min((uint32_t)h.level_min, b.z);
            h.n_silent += b.w & 0xFFFFu; h.n_clipped += b.w >> 16;
        }
        if (p.run < alarm && p.run + r.x >= alarm) p.alarms += 1u;
        p.run = r.w ? r.y : p.run + r.y;
        p.alarms += r.z;
    }
    if (open) hold[c] = h;
    if (probe) probe[c] = p;
}

// The table k_encode_lut16 keeps in LDS, evaluated once per context and lineage: tab[law << 16 | uint16(v)] = enc_uni(v).
template <int VARIANT>
__global__ __launch_bounds__(1024) void k_build_enc_table(uint8_t *__restrict__ tab)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2u * 65536u) return;
    tab[i] = (uint8_t)enc_uni<VARIANT>((int)(int16_t)(i & 0xFFFFu), enc_consts<VARIANT>((i >> 16) != 0u));
}

hipError_t launch_build_enc_table(int variant, uint8_t *tab, hipStream_t s)
{
    if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_build_enc_table<IGDSP_ENC_G191>), dim3(128), dim3(1024), 0, s, tab);
    else                           hipLaunchKernelGGL((k_build_enc_table<IGDSP_ENC_SUN16>), dim3(128), dim3(1024), 0, s, tab);
    return hipGetLastError();
}

hipError_t launch_encode(const LaunchCfg &cfg, const int16_t *pcm, const uint8_t *codec, uint32_t C, uint32_t F,
                         uint32_t n, uint8_t *out, int variant, hipStream_t s)
{
    const uint64_t n_samples = (uint64_t)C * F * n;
    if (n_samples == 0) return hipSuccess;
    const bool v8 = ((n & 7u) == 0u) && ((reinterpret_cast<uintptr_t>(pcm) & 15u) == 0u) &&
                    ((reinterpret_cast<uintptr_t>(out) & 7u) == 0u);
    const uint32_t cap = (uint32_t)cfg.compute_units * 8u;
    if (v8 && n_samples >= (1u << 25) && (n_samples >> 3) < 0xFFFF0000ull) {   // large batches: full 16-bit table, one block per CU (32-bit group ids)
        const uint32_t groups = (uint32_t)(n_samples >> 3);    // 32-bit group ids (checked above)
        const uint32_t grid = blocks_for(groups, 1024, (uint32_t)cfg.compute_units);
        if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_encode_lut16<IGDSP_ENC_G191>), dim3(grid), dim3(kEncWaves * 64), 0, s, pcm, codec, C, n, groups, out, cfg.gqueue, cfg.enc_tab);
        else                           hipLaunchKernelGGL((k_encode_lut16<IGDSP_ENC_SUN16>), dim3(grid), dim3(kEncWaves * 64), 0, s, pcm, codec, C, n, groups, out, cfg.gqueue, cfg.enc_tab);
    } else if (v8 && n_samples >= (1u << 22)) {                 // big batches: table-driven compressor, persistent blocks
        const uint64_t groups = n_samples >> 3;
        const uint32_t grid = blocks_for(groups, 1024, (uint32_t)cfg.compute_units * 2u);
        if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_encode_v8_table<IGDSP_ENC_G191>), dim3(grid), dim3(1024), 0, s, pcm, codec, C, n, groups, out);
        else                           hipLaunchKernelGGL((k_encode_v8_table<IGDSP_ENC_SUN16>), dim3(grid), dim3(1024), 0, s, pcm, codec, C, n, groups, out);
    } else if (v8) {
        const uint64_t groups = n_samples >> 3;
        const uint32_t grid = blocks_for(groups, 256, cap);
        if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_encode_v8<IGDSP_ENC_G191>), dim3(grid), dim3(256), 0, s, pcm, codec, C, n, groups, out);
        else                           hipLaunchKernelGGL((k_encode_v8<IGDSP_ENC_SUN16>), dim3(grid), dim3(256), 0, s, pcm, codec, C, n, groups, out);
    } else {
        const uint32_t grid = blocks_for(n_samples, 256, cap);
        if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_encode_scalar<IGDSP_ENC_G191>), dim3(grid), dim3(256), 0, s, pcm, codec, C, n, n_samples, out);
        else                           hipLaunchKernelGGL((k_encode_scalar<IGDSP_ENC_SUN16>), dim3(grid), dim3(256), 0, s, pcm, codec, C, n, n_samples, out);
    }
    return hipGetLastError();
}

hipError_t launch_encode_table(const LaunchCfg &cfg, const int16_t *pcm, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                               uint8_t *out, int variant, hipStream_t s)
{
    const uint64_t n_samples = (uint64_t)C * F * n;
    if (n_samples == 0) return hipSuccess;
    const uint32_t grid = blocks_for(n_samples, 1024 * 16, (uint32_t)cfg.compute_units);
    if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_encode_table<IGDSP_ENC_G191>), dim3(grid), dim3(1024), 0, s, pcm, codec, C, n, n_samples, out);
    else                           hipLaunchKernelGGL((k_encode_table<IGDSP_ENC_SUN16>), dim3(grid), dim3(1024), 0, s, pcm, codec, C, n, n_samples, out);
    return hipGetLastError();
}

// Waves per block of the block-owned round-trip kernels.  Unlike the read-heavy kernels (the more waves the better: each has one
// item of loads in flight) the 1 : 1 read / write mix is fastest with FEW resident waves per CU: at 160-byte frames 16 / 14 / 12 / 10
// / 8 / 6 / 4 / 3 waves ran 0.4613 / 0.4607 / 0.4580 / 0.4552 / 0.4522 / 0.4498 / 0.4519 / 0.4785 ms in same-box A/B builds.
static uint32_t rtb_waves(uint32_t n, int max_waves, uint32_t gpb)
{
    // about 60 KB of loads in flight per CU: 6 waves at 160 bytes per frame, 4 at 240, 10 (the strips' limit) at 80 and below; the tailed
    // sizes want two more (164 bytes: 4 / 6 / 8 / 10 waves 0.6413 / 0.5146 / 0.4854 / 0.4915 ms; 240: 0.6636 / 0.6740 / 0.6746 / 0.6804)
    uint32_t w = n >= 200u ? 4u : (n > 160u ? 8u : (n >= 120u ? 6u : (n >= 48u ? 10u : 16u)));
    if (const char *e = std::getenv("IGDSP_RTB_WAVES")) w = (uint32_t)std::max(1, std::atoi(e));   // experiments
    return std::max(std::max(gpb, 1u), std::min(w, (uint32_t)max_waves));                         // (a wave per owned group folds it at the block's end)
}

// Where the block-owned form pays (65 536 channels = 256 blocks of four groups is the tuned case; measured around it, 128 frames,
// placed buffers, static / block-owned ms): 0.6 to 1 round of blocks — 10 240 ch 0.0863 / 0.0841, 12 288 ch 0.1108 / 0.0856.  Below 0.6 of a
// round the static form spreads its units over more CUs (8 192 ch 0.0754 / 0.0804, 4 096 ch 0.0649 / 0.0769); 1.5 rounds idle half the chip
// in the second (24 576 ch 0.2003 / 0.2099, 49 152 ch 0.3572 / 0.3769), and even two whole rounds lose to the static form, whose blocks
// stay (131 072 ch 0.9216 / 0.9311).
static bool rtb_fills(uint32_t blocks, uint32_t rounds, uint32_t cus)
{
    return rounds == 1u && blocks * 10u >= cus * 6u;
}

hipError_t launch_roundtrip(const LaunchCfg &cfg, int kernel_variant, const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F,
                            uint32_t n, uint8_t *out, igdsp_frame_stats *stats, igdsp_chan_hold *hold,
                            const uint8_t *gate, int variant, hipStream_t s)
{
    if ((uint64_t)C * F == 0) return hipSuccess;
    // Item order of the fused kernels.  Consecutive groups per block pay when the output's halves lie in two memory classes
    // (0.4695 vs 0.4756 ms); with the whole output in one class it is the other way round (0.5116 vs 0.4920 ms): tools/rt_knobs.py,
    // alternating in one process.  IGDSP_RT_ORDER overrides (experiments, and the test of the order the placement would pick).
    uint32_t order = cfg.out_spread ? 1u : 0u;
    if (const char *e = std::getenv("IGDSP_RT_ORDER")) order = (uint32_t)std::atoi(e);
    uint32_t mid_start = cfg.out_spread ? 1u : 0u;               // block-owned form: odd blocks walk the frames from the middle (both halves of a spread output written at any moment)
    if (const char *e = std::getenv("IGDSP_RT_MID")) mid_start = std::atoi(e) != 0 ? 1u : 0u;   // experiments
    // The fused channel-group-major kernels take whole groups of 64 channels of 160-byte frames in 16-byte aligned
    // buffers; the C % 64 channels left over, and every other shape (n != 160, unaligned buffers), go through
    // k_roundtrip_general on the same stream.  kernel_variant 4 selects the compressor-cell-table form of the fused
    // kernel (k_roundtrip_chunk64, kept for A/B runs); the default folds the compressor into the expansion LUT.
    const bool aligned = ((reinterpret_cast<uintptr_t>(payload) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(stats)) & 15u) == 0u;
    // the reference's other frame sizes (24, 80, 164 / 168, 240; dword-aligned buffers suffice) keep the fused walk: k_roundtrip_strided
    const uint32_t Qn = n >> 4, Tn = (n >> 2) & 3u;
    const bool strided = kernel_variant != 1 && n != (uint32_t)kFrame && (n & 3u) == 0u && Tn != 3u && n >= 16u &&
                         (((reinterpret_cast<uintptr_t>(payload) | reinterpret_cast<uintptr_t>(out)) & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(stats) & 15u) == 0u) &&
                         ((Qn == 1u && Tn != 0u) || (Qn == 5u && Tn == 0u) || (Qn == 10u && Tn != 0u) || (Qn == 15u && Tn == 0u));
    if (strided && C >= (uint32_t)kSuperFrames) {
        const uint32_t n_groups_s = C / kSuperFrames;
        const uint32_t want = (uint32_t)cfg.compute_units * (uint32_t)kRtlWaves;
        uint32_t n_seg = n_groups_s >= want ? 1u : (want + n_groups_s - 1u) / n_groups_s;
        n_seg = std::max(1u, std::min(n_seg, std::max(1u, F / 8u)));
        n_seg = std::max(n_seg, F / 65535u + 1u);
        const uint32_t grid = blocks_for((uint64_t)n_groups_s * n_seg, kRtlWaves, (uint32_t)cfg.compute_units);
        // block-owned form (same rule as for 160-byte frames below)
        uint32_t gpb = 1u;
        const uint32_t cus = (uint32_t)std::max(1, cfg.compute_units);
        for (uint32_t g = 4u; g > 1u; g >>= 1) if (n_groups_s % g == 0u && n_groups_s / g >= cus) { gpb = g; break; }
        if (const char *e = std::getenv("IGDSP_RT_GPB")) { const uint32_t g = (uint32_t)std::atoi(e); if ((g == 1u || g == 2u || g == 4u) && n_groups_s % g == 0u) gpb = g; }   // tests
        const uint32_t blocks = n_groups_s / gpb, rounds = (blocks + cus - 1u) / cus;
        bool blk = F <= 65535u && rtb_fills(blocks, rounds, cus);
        if (const char *e = std::getenv("IGDSP_RT_BLK")) blk = F <= 65535u && std::atoi(e) != 0;       // experiments and tests
        if (blk) {
            const uint32_t gsh = gpb == 4u ? 2u : (gpb == 2u ? 1u : 0u);
            const dim3 gb(blocks), bb(rtb_waves(n, Qn <= 1u ? 16 : kRtsbWaves, gpb) * 64u);
#define IGDSP_RTSB(QV, TV)                                                                                                                                     \
            if (Qn == QV && (Tn != 0u) == TV) {                                                                                                                 \
                if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_roundtrip_strided<QV, TV, IGDSP_ENC_G191, true>), gb, bb, 0, s, payload, codec, C, F, n, out, stats, hold, gate, gpb, n_groups_s, gsh | (mid_start << 2));  \
                else                           hipLaunchKernelGGL((k_roundtrip_strided<QV, TV, IGDSP_ENC_SUN16, true>), gb, bb, 0, s, payload, codec, C, F, n, out, stats, hold, gate, gpb, n_groups_s, gsh | (mid_start << 2)); \
            }
            IGDSP_RTSB(1, true) IGDSP_RTSB(5, false) IGDSP_RTSB(10, true) IGDSP_RTSB(15, false)
#undef IGDSP_RTSB
        } else {
        const dim3 g3(grid), b3(kRtlWaves * 64);
#define IGDSP_RTS(QV, TV)                                                                                                                                      \
        if (Qn == QV && (Tn != 0u) == TV) {                                                                                                                     \
            if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_roundtrip_strided<QV, TV, IGDSP_ENC_G191>), g3, b3, 0, s, payload, codec, C, F, n, out, stats, hold, gate, n_seg, n_groups_s, order);  \
            else                           hipLaunchKernelGGL((k_roundtrip_strided<QV, TV, IGDSP_ENC_SUN16>), g3, b3, 0, s, payload, codec, C, F, n, out, stats, hold, gate, n_seg, n_groups_s, order); \
        }
        IGDSP_RTS(1, true) IGDSP_RTS(5, false) IGDSP_RTS(10, true) IGDSP_RTS(15, false)
#undef IGDSP_RTS
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const uint32_t c_first = n_groups_s * (uint32_t)kSuperFrames, c_count = C - c_first;
        if (c_count != 0u) {
            const uint32_t gridg = blocks_for(c_count, 4, (uint32_t)cfg.compute_units * 8u);
            if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_roundtrip_general<IGDSP_ENC_G191>), dim3(gridg), dim3(256), 0, s, payload, codec, C, F, n, c_first, c_count, out, stats, hold, gate);
            else                           hipLaunchKernelGGL((k_roundtrip_general<IGDSP_ENC_SUN16>), dim3(gridg), dim3(256), 0, s, payload, codec, C, F, n, c_first, c_count, out, stats, hold, gate);
        }
        return hipGetLastError();
    }
    const uint32_t n_groups = (n == (uint32_t)kFrame && aligned && kernel_variant != 1) ? C / kSuperFrames : 0u;
    if (n_groups != 0u) {
        // fill the chip: at least one work item per resident wave; a segment is never shorter than 8 frames
        const int waves = kernel_variant == 4 ? kRtWaves : kRtlWaves;
        const uint32_t want = (uint32_t)cfg.compute_units * (uint32_t)waves;
        uint32_t n_seg = n_groups >= want ? 1u : (want + n_groups - 1u) / n_groups;
        if (const char *e = std::getenv("IGDSP_RT_NSEG")) n_seg = (uint32_t)std::max(1, std::atoi(e));   // experiments
        n_seg = std::max(1u, std::min(n_seg, std::max(1u, F / 8u)));
        n_seg = std::max(n_seg, F / 65535u + 1u);               // the fused kernels count silent / clipped frames of a segment in 16 bits
        const uint32_t grid = blocks_for((uint64_t)n_groups * n_seg, waves, (uint32_t)cfg.compute_units);
        // block-owned form: gpb = as many groups per block (<= 4: the LDS) as still give every CU a block; taken where rtb_fills says
        // it pays and F fits the 16-bit silent / clipped counts
        uint32_t gpb = 1u;
        const uint32_t cus = (uint32_t)std::max(1, cfg.compute_units);
        for (uint32_t g = 4u; g > 1u; g >>= 1) if (n_groups % g == 0u && n_groups / g >= cus) { gpb = g; break; }
        if (const char *e = std::getenv("IGDSP_RT_GPB")) { const uint32_t g = (uint32_t)std::atoi(e); if ((g == 1u || g == 2u || g == 4u) && n_groups % g == 0u) gpb = g; }   // tests
        const uint32_t blocks = n_groups / gpb, rounds = (blocks + cus - 1u) / cus;
        bool blk = kernel_variant != 4 && F <= 65535u && rtb_fills(blocks, rounds, cus);
        if (const char *e = std::getenv("IGDSP_RT_BLK")) blk = kernel_variant != 4 && F <= 65535u && std::atoi(e) != 0;   // experiments and tests
        if (blk) {
            const uint32_t gsh = gpb == 4u ? 2u : (gpb == 2u ? 1u : 0u);
            if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_roundtrip_blk64<IGDSP_ENC_G191>), dim3(blocks), dim3(rtb_waves(n, kRtbWaves, gpb) * 64u), 0, s, payload, codec, C, F, out, stats, hold, gate, gpb, gsh, mid_start);
            else                           hipLaunchKernelGGL((k_roundtrip_blk64<IGDSP_ENC_SUN16>), dim3(blocks), dim3(rtb_waves(n, kRtbWaves, gpb) * 64u), 0, s, payload, codec, C, F, out, stats, hold, gate, gpb, gsh, mid_start);
        } else if (kernel_variant == 4) {
            if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_roundtrip_chunk64<IGDSP_ENC_G191>), dim3(grid), dim3(kRtWaves * 64), 0, s, payload, codec, C, F, out, stats, hold, gate, n_seg, n_groups);
            else                           hipLaunchKernelGGL((k_roundtrip_chunk64<IGDSP_ENC_SUN16>), dim3(grid), dim3(kRtWaves * 64), 0, s, payload, codec, C, F, out, stats, hold, gate, n_seg, n_groups);
        } else {
            if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_roundtrip_lut64<IGDSP_ENC_G191>), dim3(grid), dim3(kRtlWaves * 64), 0, s, payload, codec, C, F, out, stats, hold, gate, n_seg, n_groups, order);
            else                           hipLaunchKernelGGL((k_roundtrip_lut64<IGDSP_ENC_SUN16>), dim3(grid), dim3(kRtlWaves * 64), 0, s, payload, codec, C, F, out, stats, hold, gate, n_seg, n_groups, order);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const uint32_t c_first = n_groups * (uint32_t)kSuperFrames, c_count = C - c_first;
    if (c_count != 0u) {
        const uint32_t grid = blocks_for(c_count, 4, (uint32_t)cfg.compute_units * 8u);
        if (variant == IGDSP_ENC_G191) hipLaunchKernelGGL((k_roundtrip_general<IGDSP_ENC_G191>), dim3(grid), dim3(256), 0, s, payload, codec, C, F, n, c_first, c_count, out, stats, hold, gate);
        else                           hipLaunchKernelGGL((k_roundtrip_general<IGDSP_ENC_SUN16>), dim3(grid), dim3(256), 0, s, payload, codec, C, F, n, c_first, c_count, out, stats, hold, gate);
    }
    return hipGetLastError();
}

hipError_t launch_hold_update(const igdsp_frame_stats *stats, const uint16_t *len, uint32_t C, uint32_t F, uint32_t n,
                              igdsp_chan_hold *hold, const uint8_t *gate, hipStream_t s)
{
    if (C == 0 || F == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hold_update, dim3((C + 255) / 256), dim3(256), 0, s, stats, len, C, F, n, hold, gate);
    return hipGetLastError();
}

hipError_t launch_window_update(const igdsp_frame_stats *stats, const igdsp_rtp_info *info, const uint16_t *len, uint32_t C, uint32_t F, uint32_t n,
                                uint32_t gate_mode, uint32_t alarm, igdsp_chan_hold *hold, const uint8_t *gate, igdsp_chan_probe *probe, hipStream_t s)
{
    if (C == 0 || F == 0) return hipSuccess;
    hipLaunchKernelGGL(k_window_update, dim3((C + 255) / 256), dim3(256), 0, s, stats, info, len, C, F, n, gate_mode, alarm, hold, gate, probe);
    return hipGetLastError();
}

hipError_t launch_window_finish(const uint4 *work, uint32_t C, uint32_t n_seg, uint32_t alarm, igdsp_chan_hold *hold, const uint8_t *gate,
                                igdsp_chan_probe *probe, hipStream_t s)
{
    if (C == 0 || n_seg == 0) return hipSuccess;
    hipLaunchKernelGGL(k_window_finish, dim3((C + 255) / 256), dim3(256), 0, s, work, C, n_seg, alarm, hold, gate, probe);
    return hipGetLastError();
}

hipError_t launch_flush_fold(const igdsp_frame_stats *stA, const igdsp_frame_stats *stB, const uint16_t *lenB, const uint2 *seq, const uint2 *runs,
                             uint32_t n_channels, uint32_t gate_mode, uint32_t alarm, igdsp_chan_hold *hold, igdsp_chan_probe *probe,
                             igdsp_frame_stats *last, hipStream_t s)
{
    if (n_channels == 0) return hipSuccess;
    hipLaunchKernelGGL(k_flush_fold, dim3((n_channels + 255) / 256), dim3(256), 0, s, stA, stB, lenB, seq, runs, n_channels, gate_mode, alarm, hold, probe, last);
    return hipGetLastError();
}

hipError_t launch_hold_reset(igdsp_chan_hold *hold, uint32_t C, const uint8_t *mask, hipStream_t s)
{
    if (C == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hold_reset, dim3((C + 255) / 256), dim3(256), 0, s, hold, C, mask);
    return hipGetLastError();
}


}  // namespace igdsp
