// igdsp_k_misc.hip — recorder file images (k_wav_expand*), G.726 reorder, the synthetic generator, stream calibration kernels.
// Hand-written gfx950 (CDNA4, wave64) kernels; no MFMA: the path is a byte stream with ~4 integer ops per sample, bounded by
// HBM.  Shared device code: igdsp_device.h.
#include "igdsp_device.h"

namespace igdsp {

// ============================================================================
// SURVEY 8(f) rank 2 on the device — recorder-compatible output (WavWriter.cpp:63-156).  For every channel c one
// complete WavWriter file image: the 44-byte header WavWriter::start writes (tag 7, "2 channels", 16 bit, rate, rate * 4,
// align 4) with the two sizes WavWriter::stop patches in, then every payload byte b as the two bytes [b, 0x00]
// (write_little_endian with the channel count used as the byte count).  A [F][C][n] -> [C][44 + 2 F n] transposition
// with a 1 : 2 byte expansion: a block stages a tile of 16 channels x 16 frames in LDS (reads: 16 n contiguous bytes per
// frame row), then each wave streams whole channels out, 1 KiB of contiguous file bytes per store instruction.
// ============================================================================
constexpr int kWavTile = 16;

__device__ __forceinline__ void wav_header_words(uint32_t (&h)[11], uint32_t rate, uint32_t data_bytes)
{
    h[0] = 0x46464952u;                  // "RIFF"
    h[1] = 36u + data_bytes;
    h[2] = 0x45564157u;                  // "WAVE"
    h[3] = 0x20746D66u;                  // "fmt "
    h[4] = 16u;
    h[5] = 0x0007u | (2u << 16);         // format tag 7, "channels" 2
    h[6] = rate;
    h[7] = rate * 4u;
    h[8] = 4u | (16u << 16);             // block align 4, 16 bits per sample
    h[9] = 0x61746164u;                  // "data"
    h[10] = data_bytes;
}

typedef uint32_t u32x4_st4_t __attribute__((ext_vector_type(4), aligned(4)));

// n % 8 == 0: output pieces of 16 bytes (8 payload bytes) never straddle a frame
__global__ __launch_bounds__(256) void k_wav_expand16(const uint8_t *__restrict__ payload, uint32_t C, uint32_t F, uint32_t n,
                                                      uint32_t rate, uint8_t *__restrict__ files, uint64_t file_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t wav_tile[];     // [frame row][16 n]
    // channel tiles are visited alternately from the two halves of the channel range: the file images of the low and the high
    // channels are the two halves of the output buffer, which igdsp_io_alloc puts into two different classes of device memory
    const uint32_t c0 = spread_batch(blockIdx.x, gridDim.x) * (uint32_t)kWavTile, f0 = blockIdx.y * (uint32_t)kWavTile;
    const uint32_t nc = min((uint32_t)kWavTile, C - c0), nf = min((uint32_t)kWavTile, F - f0);
    const uint32_t row_bytes = nc * n;                                    // bytes of this tile in one frame row (multiple of 8)
    const uint32_t row_lds = (uint32_t)kWavTile * n;
    // read: rows of nc * n contiguous bytes, 16 bytes per lane at dword alignment (the last piece of a row may be 8 bytes)
    const uint32_t ppr = (row_bytes + 15u) >> 4;
    for (uint32_t p = threadIdx.x; p < nf * ppr; p += blockDim.x) {
        const uint32_t r = p / ppr, k = p - r * ppr;
        const uint8_t *src = payload + ((uint64_t)(f0 + r) * C + c0) * n + 16u * k;
        uint4 v;
        if (16u * k + 16u <= row_bytes) v = ld16_dw(src);
        else { const uint2 t = *reinterpret_cast<const uint2 *>(src); v = make_uint4(t.x, t.y, 0u, 0u); }
        *reinterpret_cast<uint4 *>(wav_tile + r * row_lds + 16u * k) = v;
    }
    __syncthreads();
    // write: channel by channel, consecutive lanes = consecutive 16-byte pieces of the file
    const uint32_t ppc = nf * n / 8u;                                     // output pieces per channel in this tile
    const uint32_t data_bytes = 2u * F * n;
    for (uint32_t p = threadIdx.x; p < nc * ppc; p += blockDim.x) {
        const uint32_t c = p / ppc, j = p - c * ppc;
        const uint32_t b = 8u * j, r = b / n, i = b - r * n;
        const uint2 t = *reinterpret_cast<const uint2 *>(wav_tile + r * row_lds + c * n + i);
        u32x4_st4_t o;
        o.x = __builtin_amdgcn_perm(0u, t.x, 0x0C010C00u); o.y = __builtin_amdgcn_perm(0u, t.x, 0x0C030C02u);
        o.z = __builtin_amdgcn_perm(0u, t.y, 0x0C010C00u); o.w = __builtin_amdgcn_perm(0u, t.y, 0x0C030C02u);
        uint8_t *dst = files + (uint64_t)(c0 + c) * file_stride + 44u + 2ull * ((uint64_t)f0 * n + b);
        *reinterpret_cast<u32x4_st4_t *>(dst) = o;
    }
    if (blockIdx.y == 0 && threadIdx.x < nc * 11u) {                      // the tile of the first frames also writes the headers
        uint32_t h[11];
        wav_header_words(h, rate, data_bytes);
        const uint32_t c = threadIdx.x / 11u, w = threadIdx.x - c * 11u;
        uint32_t v = h[0];
#pragma unroll
        for (int q = 1; q < 11; ++q) v = (w == (uint32_t)q) ? h[q] : v;
        reinterpret_cast<uint32_t *>(files + (uint64_t)(c0 + c) * file_stride)[w] = v;
    }
}

// any n, any alignment: one thread per payload byte
__global__ __launch_bounds__(256) void k_wav_expand_bytes(const uint8_t *__restrict__ payload, uint32_t C, uint32_t F, uint32_t n,
                                                          uint32_t rate, uint8_t *__restrict__ files, uint64_t file_stride)
{
    const uint64_t per_ch = (uint64_t)F * n, total = per_ch * C;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(g / per_ch);
        const uint64_t k = g - (uint64_t)c * per_ch;                      // payload byte k of channel c: frame k / n, byte k % n
        const uint32_t f = (uint32_t)(k / n), i = (uint32_t)(k - (uint64_t)f * n);
        const uint8_t b = payload[((uint64_t)f * C + c) * n + i];
        uint8_t *dst = files + (uint64_t)c * file_stride + 44u + 2ull * k;
        dst[0] = b; dst[1] = 0;
    }
    const uint32_t data_bytes = 2u * F * n;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < C * 44u; t += gridDim.x * blockDim.x) {
        uint32_t h[11];
        wav_header_words(h, rate, data_bytes);
        const uint32_t c = t / 44u, o = t - c * 44u;
        uint32_t v = h[0];
#pragma unroll
        for (int q = 1; q < 11; ++q) v = ((o >> 2) == (uint32_t)q) ? h[q] : v;
        files[(uint64_t)c * file_stride + o] = (uint8_t)(v >> (8u * (o & 3u)));
    }
}

hipError_t launch_wav_expand(const LaunchCfg &cfg, const uint8_t *payload, uint32_t C, uint32_t F, uint32_t n, uint32_t rate,
                             uint8_t *files, uint64_t file_stride, hipStream_t s)
{
    if ((uint64_t)C * F == 0) return hipSuccess;
    const bool fast = (n & 7u) == 0u && (file_stride & 3u) == 0u && ((reinterpret_cast<uintptr_t>(payload) | reinterpret_cast<uintptr_t>(files)) & 3u) == 0u &&
                      (F + kWavTile - 1) / kWavTile <= 65535u;
    if (fast) {
        const dim3 grid((C + kWavTile - 1) / kWavTile, (F + kWavTile - 1) / kWavTile);
        hipLaunchKernelGGL(k_wav_expand16, grid, dim3(256), (size_t)kWavTile * kWavTile * n, s, payload, C, F, n, rate, files, file_stride);
    } else {
        const uint64_t total = (uint64_t)C * F * n;
        hipLaunchKernelGGL(k_wav_expand_bytes, dim3(blocks_for(total, 256, (uint32_t)cfg.compute_units * 16u)), dim3(256), 0, s, payload, C, F, n, rate, files, file_stride);
    }
    return hipGetLastError();
}

// ============================================================================
// SURVEY 8(f) rank 4 — G.726 code-word reorder (changeUplinkOrder, roip_ed137.cpp:6379-6499), byte-parallel.
// Modes 1 / 3 permute bit fields inside each byte: whole dwords with masks, 16 B per lane.
// Modes 2 / 4 permute inside 3- / 5-byte groups: one lane takes four groups (12 / 20 bytes = 3 / 5 aligned dwords).
// ============================================================================
__device__ __forceinline__ uint32_t g726_w2(uint32_t w)   // reverse the four 2-bit fields of each byte
{
    return ((w & 0x03030303u) << 6) | ((w & 0x0C0C0C0Cu) << 2) | ((w & 0x30303030u) >> 2) | ((w & 0xC0C0C0C0u) >> 6);
}
__device__ __forceinline__ uint32_t g726_w4(uint32_t w) { return ((w >> 4) & 0x0F0F0F0Fu) | ((w << 4) & 0xF0F0F0F0u); }

__device__ __forceinline__ uint32_t g726_g3(uint32_t V)   // 24-bit group, reference field layout
{
    const uint32_t S1 = V & 7u, S2 = (V >> 3) & 7u, S3 = (V >> 7) & 3u, S3_ = (V >> 6) & 1u, S4 = (V >> 9) & 7u;
    const uint32_t S5 = (V >> 12) & 7u, S6 = (V >> 17) & 1u, S6_ = (V >> 15) & 3u, S7 = (V >> 18) & 7u, S8 = (V >> 21) & 7u;
    return (S3 | (S2 << 2) | (S1 << 5)) | ((S6 | (S5 << 1) | (S4 << 4) | (S3_ << 7)) << 8) | ((S8 | (S7 << 3) | (S6_ << 6)) << 16);
}

__device__ __forceinline__ void g726_g5(const uint32_t t0, const uint32_t t1, const uint32_t t2, const uint32_t t3, const uint32_t t4,
                                        uint32_t (&o)[5])
{
    const uint32_t S1 = t0 & 0x1Fu, S2 = ((t1 << 1) | (t0 >> 7)) & 7u;      // S2_ (2-bit field <- 0 or 4) is always 0 in the reference
    const uint32_t S3 = (t1 >> 2) & 0x1Fu, S4 = (t2 >> 3) & 1u, S4_ = ((t2 << 1) | (t1 >> 7)) & 0x0Fu;
    const uint32_t S5 = ((t3 << 3) | (t2 >> 5)) & 0x0Fu, S5_ = (t2 >> 4) & 1u, S6 = (t3 >> 1) & 0x1Fu;
    const uint32_t S7 = (t4 >> 1) & 3u, S7_ = ((t4 << 2) | (t3 >> 6)) & 7u, S8 = (t4 >> 3) & 0x1Fu;
    o[0] = S2 | (S1 << 3); o[1] = S4 | (S3 << 1); o[2] = S5 | (S4_ << 4); o[3] = S7 | (S6 << 2) | (S5_ << 7); o[4] = S8 | (S7_ << 5);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_g726_bytes(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t n_units)
{
    // unit = 16 bytes (modes 1, 3), 12 bytes (mode 2), 20 bytes (mode 4); all dword aligned
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += (uint64_t)gridDim.x * blockDim.x) {
        if (MODE == 1 || MODE == 3) {
            const uint4 d = ld_stream(reinterpret_cast<const uint4 *>(in) + u);
            uint4 r;
            if (MODE == 1) r = make_uint4(g726_w2(d.x), g726_w2(d.y), g726_w2(d.z), g726_w2(d.w));
            else r = make_uint4(g726_w4(d.x), g726_w4(d.y), g726_w4(d.z), g726_w4(d.w));
            reinterpret_cast<uint4 *>(out)[u] = r;
        } else if (MODE == 2) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(in) + u * 3u;
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            const uint32_t r0 = g726_g3(d0 & 0xFFFFFFu), r1 = g726_g3((d0 >> 24) | ((d1 & 0xFFFFu) << 8));
            const uint32_t r2 = g726_g3((d1 >> 16) | ((d2 & 0xFFu) << 16)), r3 = g726_g3(d2 >> 8);
            uint32_t *q = reinterpret_cast<uint32_t *>(out) + u * 3u;
            q[0] = r0 | (r1 << 24); q[1] = (r1 >> 8) | (r2 << 16); q[2] = (r2 >> 16) | (r3 << 8);
        } else {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(in) + u * 5u;
            uint32_t d[5], b[20], r[20];
#pragma unroll
            for (int i = 0; i < 5; ++i) d[i] = p[i];
#pragma unroll
            for (int i = 0; i < 20; ++i) b[i] = (d[i >> 2] >> (8 * (i & 3))) & 0xFFu;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t o[5];
                g726_g5(b[5 * g], b[5 * g + 1], b[5 * g + 2], b[5 * g + 3], b[5 * g + 4], o);
#pragma unroll
                for (int i = 0; i < 5; ++i) r[5 * g + i] = o[i];
            }
            uint32_t *q = reinterpret_cast<uint32_t *>(out) + u * 5u;
#pragma unroll
            for (int i = 0; i < 5; ++i) q[i] = r[4 * i] | (r[4 * i + 1] << 8) | (r[4 * i + 2] << 16) | (r[4 * i + 3] << 24);
        }
    }
}

// tail / unaligned: one lane per group of 1, 3 or 5 bytes
template <int MODE>
__global__ __launch_bounds__(256) void k_g726_groups(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t first_byte,
                                                     uint64_t n_groups)
{
    constexpr uint32_t G = (MODE == 2) ? 3u : (MODE == 4 ? 5u : 1u);
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *p = in + first_byte + g * G;
        uint8_t *q = out + first_byte + g * G;
        if (MODE == 1) q[0] = (uint8_t)g726_w2(p[0]);
        else if (MODE == 3) q[0] = (uint8_t)g726_w4(p[0]);
        else if (MODE == 2) {
            const uint32_t r = g726_g3((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16));
            q[0] = (uint8_t)r; q[1] = (uint8_t)(r >> 8); q[2] = (uint8_t)(r >> 16);
        } else {
            uint32_t o[5];
            g726_g5(p[0], p[1], p[2], p[3], p[4], o);
#pragma unroll
            for (int i = 0; i < 5; ++i) q[i] = (uint8_t)o[i];
        }
    }
}

template <int MODE>
static hipError_t launch_g726_mode(const LaunchCfg &cfg, const uint8_t *in, uint8_t *out, uint64_t n_bytes, hipStream_t s)
{
    constexpr uint64_t G = (MODE == 2) ? 3 : (MODE == 4 ? 5 : 1);
    constexpr uint64_t U = (MODE == 2) ? 12 : (MODE == 4 ? 20 : 16);
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0u;
    const uint64_t units = aligned ? n_bytes / U : 0;
    const uint32_t cap = (uint32_t)cfg.compute_units * 8u;
    if (units) hipLaunchKernelGGL((k_g726_bytes<MODE>), dim3(blocks_for(units, 256, cap)), dim3(256), 0, s, in, out, units);
    const uint64_t rest = n_bytes - units * U;
    if (rest) hipLaunchKernelGGL((k_g726_groups<MODE>), dim3(blocks_for(rest / G, 256, cap)), dim3(256), 0, s, in, out, units * U, rest / G);
    return hipGetLastError();
}

hipError_t launch_g726(const LaunchCfg &cfg, const uint8_t *in, uint8_t *out, uint64_t n_bytes, int mode, hipStream_t s)
{
    if (n_bytes == 0) return hipSuccess;
    switch (mode) {
    case 1: return launch_g726_mode<1>(cfg, in, out, n_bytes, s);
    case 2: return launch_g726_mode<2>(cfg, in, out, n_bytes, s);
    case 3: return launch_g726_mode<3>(cfg, in, out, n_bytes, s);
    default: return launch_g726_mode<4>(cfg, in, out, n_bytes, s);
    }
}

// ============================================================================
// Synthetic D-uniform generator: 8 bytes per splitmix64 word.
// ============================================================================
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_gen_uniform(uint8_t *__restrict__ out, uint64_t n_bytes, uint64_t seed,
                                                     uint64_t first_byte)
{
    // thread handles one aligned 8-byte word of the GLOBAL stream; edges are byte-masked
    const uint64_t w0 = first_byte >> 3;
    const uint64_t n_words = ((first_byte + n_bytes + 7u) >> 3) - w0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t word = splitmix64(seed + w0 + i);
        const uint64_t g0 = (w0 + i) << 3;
        if (g0 >= first_byte && g0 + 8u <= first_byte + n_bytes && (((uintptr_t)(out + (g0 - first_byte))) & 7u) == 0u) {
            *reinterpret_cast<uint64_t *>(out + (g0 - first_byte)) = word;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                const uint64_t g = g0 + k;
                if (g >= first_byte && g < first_byte + n_bytes) out[g - first_byte] = (uint8_t)(word >> (8u * k));
            }
        }
    }
}

// Read-only stream calibration: same persistent geometry and load shape as chunk32.
__global__ __launch_bounds__(kBlockThreads) void k_stream_read(const uint4 *__restrict__ src, uint64_t n16,
                                                               uint64_t *__restrict__ sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 4u * stride < n16; i += 5u * stride) {
        const uint4 a = ld_stream(src + i), b = ld_stream(src + i + stride),
                    c = ld_stream(src + i + 2u * stride), d = ld_stream(src + i + 3u * stride),
                    e = ld_stream(src + i + 4u * stride);
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x ^ e.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y ^ e.y;
        acc.z ^= a.z ^ b.z ^ c.z ^ d.z ^ e.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w ^ e.w;
    }
    for (; i < n16; i += stride) {
        const uint4 a = ld_stream(src + i);
        acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
    }
    uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v ^= (uint32_t)__shfl_xor((int)v, m, 64);
    if ((threadIdx.x & 63u) == 0u && v == 0x9E3779B9u) atomicAdd((unsigned long long *)sink, 1ull);   // keeps the loads live
}

// Calibration of the meter kernel's full traffic pattern with nothing else: every wave reads 10 KiB super-chunks
// (ten 1 KiB loads, 16 adjacent super-chunks per block) and stores one 1 KiB record block per super-chunk —
// what a perfect implementation of the same bytes in / bytes out would take on this memory system.
__global__ __launch_bounds__(kBlockThreads) void k_stream_rw(const uint4 *__restrict__ src, uint32_t n_super, uint4 *__restrict__ dst)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t b = blockIdx.x; b * kWavesPerBlock + wave < n_super; b += gridDim.x) {
        const uint32_t sidx = b * kWavesPerBlock + wave;
        const uint4 *p = src + ((uint64_t)sidx * 640u + lane);
        uint4 v[10], acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = ld_stream(p + j * 64);
#pragma unroll
        for (int j = 0; j < 10; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        dst[(uint64_t)sidx * 64u + lane] = acc;
    }
}

// Calibration of read : write mixes: every wave reads R and writes W contiguous 1 KiB pieces per item (16 neighbouring
// items per block), nothing else.  <0,W> is a pure write stream, <R,R> a copy, <10,1> the meter's mix.
template <int R, int W>
__global__ __launch_bounds__(kBlockThreads) void k_stream_mix(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n_items, uint4 *dst2, const uint4 *src2)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const uint32_t n_real = n_items & 0x7FFFFFFFu;      // bit 31 of n_items selects the half-line write pattern
    for (uint32_t b = blockIdx.x; b * wpb + wave < n_real; b += gridDim.x) {
        const uint32_t item = b * wpb + wave;
        uint4 acc = make_uint4(item, lane, 0u, 0u);
        if (R > 0) {
            const uint4 *p = ((src2 != nullptr && (item & 1u)) ? src2 : src) + ((uint64_t)item * (uint32_t)(R * 64) + lane);
            uint4 v[R > 0 ? R : 1];
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = ld_stream(p + j * 64);
#pragma unroll
            for (int j = 0; j < R; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        }
        // dst2 != nullptr: odd items write to the second window (calibration of writes spread over two memory classes)
        uint4 *q = ((dst2 != nullptr && (item & 1u)) ? dst2 : dst) + ((uint64_t)item * (uint32_t)(W * 64) + lane);
        if (n_items & 0x80000000u) {
            // calibration of the PCM-store write pattern: a pair of store instructions fills 2 KiB, each instruction writing
            // 64-byte segments at 128-byte stride (the quad-regrouped stores of process_half<true>)
            uint4 *qb = q - lane;
#pragma unroll
            for (int j = 0; j < W; ++j) qb[(j >> 1) * 128 + (lane >> 2) * 8 + (j & 1) * 4 + (lane & 3)] = acc;
        } else {
#pragma unroll
            for (int j = 0; j < W; ++j) q[j * 64] = acc;
        }
    }
}

// Calibration of CLUSTERED record stores (round-3 question: does the same-class read / write penalty shrink when reads and writes
// alternate less often?): the meter's 10 : 1 traffic, but a wave reads K consecutive super-chunks (K x 10 KiB) before it stores
// their K record blocks as one K KiB run.  K = 1 is k_stream_rw's pattern.
template <int K>
__global__ __launch_bounds__(kBlockThreads) void k_stream_cluster(const uint4 *__restrict__ src, uint32_t n_super, uint4 *__restrict__ dst)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_items = n_super / (uint32_t)K;
    for (uint32_t b = blockIdx.x; b * kWavesPerBlock + wave < n_items; b += gridDim.x) {
        const uint32_t item = b * kWavesPerBlock + wave;
        uint4 acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint4 *p = src + (((uint64_t)item * (uint32_t)K + (uint32_t)k) * 640u + lane);
            uint4 v[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = ld_stream(p + j * 64);
            acc[k] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 10; ++j) { acc[k].x ^= v[j].x; acc[k].y ^= v[j].y; acc[k].z ^= v[j].z; acc[k].w ^= v[j].w; }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) dst[((uint64_t)item * (uint32_t)K + (uint32_t)k) * 64u + lane] = acc[k];
    }
}

hipError_t launch_stream_cluster(const LaunchCfg &cfg, const void *src, size_t bytes, void *dst, int k, hipStream_t s)
{
    const uint32_t n_super = (uint32_t)(bytes / 10240u);
    const dim3 g(cfg.compute_units), b(kBlockThreads);
    const uint4 *sp = reinterpret_cast<const uint4 *>(src);
    uint4 *dp = reinterpret_cast<uint4 *>(dst);
    if (k == 1) hipLaunchKernelGGL((k_stream_cluster<1>), g, b, 0, s, sp, n_super, dp);
    else if (k == 2) hipLaunchKernelGGL((k_stream_cluster<2>), g, b, 0, s, sp, n_super, dp);
    else if (k == 4) hipLaunchKernelGGL((k_stream_cluster<4>), g, b, 0, s, sp, n_super, dp);
    else if (k == 8) hipLaunchKernelGGL((k_stream_cluster<8>), g, b, 0, s, sp, n_super, dp);
    else if (k == 16) hipLaunchKernelGGL((k_stream_cluster<16>), g, b, 0, s, sp, n_super, dp);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_gen_uniform(uint8_t *out, uint64_t n_bytes, uint64_t seed, uint64_t first_byte, hipStream_t s)
{
    if (n_bytes == 0) return hipSuccess;
    const uint64_t words = (n_bytes >> 3) + 2;
    hipLaunchKernelGGL(k_gen_uniform, dim3(blocks_for(words, 256, 8192)), dim3(256), 0, s, out, n_bytes, seed, first_byte);
    return hipGetLastError();
}

hipError_t launch_stream_rw(const LaunchCfg &cfg, const void *src, size_t bytes, void *dst, hipStream_t s)
{
    const uint32_t n_super = (uint32_t)(bytes / 10240u);
    if (n_super == 0) return hipSuccess;
    hipLaunchKernelGGL(k_stream_rw, dim3(cfg.compute_units), dim3(kBlockThreads), 0, s, reinterpret_cast<const uint4 *>(src), n_super,
                       reinterpret_cast<uint4 *>(dst));
    return hipGetLastError();
}

hipError_t launch_stream_mix(const LaunchCfg &cfg, const void *src, void *dst, uint32_t n_items, int r, int w, int waves, hipStream_t s, void *dst2, const void *src2)
{
    if (waves < 1 || waves > 16) return hipErrorInvalidValue;
    const dim3 g(cfg.compute_units), b(waves * 64);
    const uint4 *sp = reinterpret_cast<const uint4 *>(src);
    uint4 *dp = reinterpret_cast<uint4 *>(dst);
#define IGDSP_MIX(R, W) if (r == R && w == W) { hipLaunchKernelGGL((k_stream_mix<R, W>), g, b, 0, s, sp, dp, n_items, reinterpret_cast<uint4 *>(dst2), reinterpret_cast<const uint4 *>(src2)); return hipGetLastError(); }
    IGDSP_MIX(0, 8) IGDSP_MIX(8, 8) IGDSP_MIX(8, 4) IGDSP_MIX(4, 8) IGDSP_MIX(10, 1) IGDSP_MIX(10, 0) IGDSP_MIX(8, 1) IGDSP_MIX(8, 2) IGDSP_MIX(20, 2) IGDSP_MIX(5, 1)
#undef IGDSP_MIX
    return hipErrorInvalidValue;
}

// Calibration of the PACKED packet kernels' access pattern, no per-sample work: item = 64 packets at `stride` bytes; every lane
// fetches `pieces` dword-aligned 16-byte pieces per 64-lane row exactly where k_meter_rtp64<packed> / k_meter_strided fetch theirs
// (mode 0: 12 pieces per packet = bytes 0-15, 4-19, then hdr + 16 q; mode 1: QP pieces per frame = 16 q, the last one at n - 16)
// and stores one 1 KiB record block (+ 512 B of info in mode 0) per item.
template <int ROWS>
__global__ __launch_bounds__(768) void k_stream_pieces(const uint8_t *__restrict__ src, uint32_t n_items, uint32_t stride, uint32_t hdr, int mode,
                                                        uint4 *__restrict__ dst, uint2 *__restrict__ dst2)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    uint32_t po[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane, f = p / (uint32_t)ROWS, q = p - f * (uint32_t)ROWS;
        if (mode == 0) po[j] = f * stride + (q == 0u ? 0u : (q == 1u ? 4u : hdr + 16u * (q - 2u)));
        else po[j] = f * stride + ((q == (uint32_t)ROWS - 1u && (stride & 15u)) ? stride - 16u : 16u * q);
    }
    for (uint32_t b = blockIdx.x; b * wpb + wave < n_items; b += gridDim.x) {
        const uint32_t item = b * wpb + wave;
        const uint8_t *base = src + (uint64_t)item * 64u * stride;
        uint4 v[ROWS], acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < ROWS; ++j) v[j] = ld16_dw(base + po[j]);
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        dst[(uint64_t)item * 64u + lane] = acc;
        if (dst2 != nullptr) dst2[(uint64_t)item * 64u + lane] = make_uint2(acc.x, acc.y);
    }
}

// The same compute-free packet-piece stream in the channel-group-major order of k_meter_rtp64<WIN>: a wave takes one unit =
// (group g of `groups`, segment of the F = n_items / groups frames) and walks its frames, item f * groups + g; the loads of the
// next item are issued before the current one is consumed (one item of lookahead, as the product kernel keeps).  Calibration only:
// what the walk itself costs the memory system against the ascending order of k_stream_pieces.
__global__ __launch_bounds__(768) void k_stream_walk(const uint8_t *__restrict__ src, uint32_t n_items, uint32_t stride, uint32_t hdr,
                                                      uint32_t groups, uint32_t n_seg, uint32_t trickle, uint4 *__restrict__ dst, uint2 *__restrict__ dst2)
{
    // groups == 0: ascending order instead (wave k of the grid takes items k, k + waves, ...), same one-item lookahead.
    // trickle != 0: the twelve loads of the next item go out one by one with s_sleep(trickle) between them, the way the product
    // kernels re-load a piece register the moment it has been folded, instead of back to back.
    constexpr int ROWS = 12;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    uint32_t po[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane, f = p / (uint32_t)ROWS, q = p - f * (uint32_t)ROWS;
        po[j] = f * stride + (q == 0u ? 0u : (q == 1u ? 4u : hdr + 16u * (q - 2u)));
    }
    const uint32_t total = gridDim.x * wpb;
    const bool asc = groups == 0u;
    if (asc) { groups = 1u; n_seg = 1u; }
    const uint32_t F = n_items / groups;
    for (uint32_t unit = asc ? blockIdx.x * wpb + wave : wave * gridDim.x + blockIdx.x; unit < (asc ? total : groups * n_seg); unit += total) {
        const uint32_t seg = asc ? 0u : unit / groups, g = asc ? unit : unit - seg * groups, step = asc ? total : groups;
        const uint32_t i_lo = asc ? g : (uint32_t)(((uint64_t)F * seg) / n_seg) * groups + g;
        const uint32_t i_hi = asc ? n_items : (uint32_t)(((uint64_t)F * (seg + 1u)) / n_seg) * groups;
        if (i_lo >= i_hi) continue;
        uint4 v[ROWS];
        {
            const uint8_t *base = src + (uint64_t)i_lo * 64u * stride;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) v[j] = ld16_dw(base + po[j]);
        }
        for (uint32_t item = i_lo; item < i_hi; item += step) {
            const uint32_t nxt = item + step < i_hi ? item + step : item;
            const uint8_t *nb = src + (uint64_t)nxt * 64u * stride;
            uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w;
                v[j] = ld16_dw(nb + po[j]);
                if (trickle == 1u) __builtin_amdgcn_s_sleep(4);
                if (trickle == 2u) __builtin_amdgcn_s_sleep(16);
                if (trickle == 3u) __builtin_amdgcn_s_sleep(48);
            }
            dst[(uint64_t)item * 64u + lane] = acc;
            if (dst2 != nullptr) dst2[(uint64_t)item * 64u + lane] = make_uint2(acc.x, acc.y);
        }
    }
}

hipError_t launch_stream_walk(const LaunchCfg &cfg, const void *src, uint32_t n_items, uint32_t stride, uint32_t hdr, uint32_t groups, uint32_t n_seg,
                              uint32_t trickle, void *dst, void *dst2, hipStream_t s)
{
    hipLaunchKernelGGL(k_stream_walk, dim3(cfg.compute_units), dim3(768), 0, s, reinterpret_cast<const uint8_t *>(src), n_items, stride, hdr, groups, n_seg, trickle,
                       reinterpret_cast<uint4 *>(dst), reinterpret_cast<uint2 *>(dst2));
    return hipGetLastError();
}

hipError_t launch_stream_pieces(const LaunchCfg &cfg, const void *src, uint32_t n_items, uint32_t stride, uint32_t hdr, int mode, int rows, void *dst, void *dst2, hipStream_t s)
{
    const dim3 g(cfg.compute_units), b(768);
    const uint8_t *sp = reinterpret_cast<const uint8_t *>(src);
#define IGDSP_PCS(R) if (rows == R) { hipLaunchKernelGGL((k_stream_pieces<R>), g, b, 0, s, sp, n_items, stride, hdr, mode, reinterpret_cast<uint4 *>(dst), reinterpret_cast<uint2 *>(dst2)); return hipGetLastError(); }
    IGDSP_PCS(10) IGDSP_PCS(11) IGDSP_PCS(12)
#undef IGDSP_PCS
    return hipErrorInvalidValue;
}

hipError_t launch_stream_read(const LaunchCfg &cfg, const void *src, size_t bytes, uint64_t *sink, hipStream_t s)
{
    if (bytes < 16) return hipSuccess;
    hipLaunchKernelGGL(k_stream_read, dim3(cfg.compute_units), dim3(kBlockThreads), 0, s,
                       reinterpret_cast<const uint4 *>(src), (uint64_t)(bytes >> 4), sink);
    return hipGetLastError();
}


}  // namespace igdsp
