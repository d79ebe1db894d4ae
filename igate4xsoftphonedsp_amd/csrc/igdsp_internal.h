// Internal launcher declarations shared by the kernel translation units (igdsp_k_*.hip) and igdsp_capi.hip / igdsp_io.hip.
// Not part of the ABI (include/igdsp.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "igdsp.h"

namespace igdsp {

// Geometry of the tuned n == 160 path: a wavefront owns a super-chunk of 64 consecutive channel-frames
// (10 240 contiguous bytes), processed as two halves of 32 frames = 5120 bytes = 5 wave-wide 16 B/lane loads each.
constexpr int kFrame = IGDSP_SAMPLES_PER_FRAME;        // 160 B
constexpr int kChunkFrames = 32;
constexpr int kChunkBytes = kChunkFrames * kFrame;     // 5120
constexpr int kPiecesPerFrame = kFrame / 16;           // 10 x 16 B
constexpr int kPiecesPerChunk = kChunkFrames * kPiecesPerFrame;  // 320
constexpr int kLoadsPerChunk = kPiecesPerChunk / 64;   // 5
constexpr int kWavesPerBlock = 16;                     // 1024 threads, one block per CU
constexpr int kBlockThreads = kWavesPerBlock * 64;

struct LaunchCfg {
    int compute_units;   // persistent grid = compute_units blocks
    uint32_t *gqueue;    // device-wide work counter {next batch, blocks done} for this launch, zero on entry
                         // and re-armed by the kernel itself; nullptr = static per-block distribution
    bool out_spread = false;   // the launch's bulk output sits half in one, half in another memory class (igdsp_io_alloc)
    const uint8_t *enc_tab = nullptr;   // igdsp_encode: the context's ready-made compressor table of this lineage (2 x 65 536 bytes), or nullptr
};

// The ED-137 gated window of a fused packet launch (igdsp_decode_meter_window): work = uint4[n_seg][3][C] unit summaries (window
// words 0, 1 and the silence-run word, k_window_finish); a unit = (one of n_groups = C / 64 channel groups, one of n_seg segments
// of the F frames).
struct WinArgs {
    uint4 *work = nullptr;
    uint32_t gate_mask = 0, alarm = IGDSP_PROBE_ALARM, n_seg = 1, n_groups = 0, F = 0;   // gate_mask: ED-137 bits that open the frame gate (0 = always)
    // block-owned form (gpb != 0): a block owns gpb = 1 << gsh consecutive channel groups for the launch, their windows live in its LDS
    // and it folds them into hold / probe itself; work = uint4[n_groups / gpb][F][gpb] probe / reset masks
    uint32_t gpb = 0, gsh = 0;
    igdsp_chan_hold *hold = nullptr;
    const uint8_t *gate = nullptr;
    igdsp_chan_probe *probe = nullptr;
};
constexpr int kWinRing = 16;                             // frames of a group that may be folded before an earlier one is (power of two)
constexpr int kWinBlkCh = 256;                           // channels a block can own (7 dwords of LDS each)
constexpr uint32_t kWinMaxSeg = 8;                       // igdsp_window_work_bytes = kWinMaxSeg x C x 48

hipError_t init_device_attributes();       // per-device kernel attributes; igdsp_create calls it with its device current
hipError_t launch_decode_meter(const LaunchCfg &cfg, int variant,
                               const uint8_t *payload, const uint8_t *codec, const uint16_t *len,
                               uint32_t C, uint32_t F, uint32_t n,
                               igdsp_frame_stats *stats, int16_t *pcm,
                               igdsp_aggregate *agg, uint32_t rank, hipStream_t s);
hipError_t launch_decode_meter_rtp(const LaunchCfg &cfg, const uint8_t *slots, const uint16_t *sizes, const uint8_t *codec, uint32_t C,
                                   uint32_t F, uint32_t stride, uint32_t hdr, igdsp_frame_stats *stats, igdsp_rtp_info *info,
                                   igdsp_aggregate *agg, uint32_t rank, hipStream_t s, const uint8_t *radio = nullptr, const WinArgs *win = nullptr);
// window fold of records (per-frame ED-137 gates, silence run): igdsp_window_update; and the chaining of a fused launch's per-segment
// run summaries into probe[c]
hipError_t launch_window_update(const igdsp_frame_stats *stats, const igdsp_rtp_info *info, const uint16_t *len, uint32_t C, uint32_t F, uint32_t n,
                                uint32_t gate_mode, uint32_t alarm, igdsp_chan_hold *hold, const uint8_t *gate, igdsp_chan_probe *probe, hipStream_t s);
hipError_t launch_window_finish(const uint4 *work, uint32_t C, uint32_t n_seg, uint32_t alarm, igdsp_chan_hold *hold, const uint8_t *gate,
                                igdsp_chan_probe *probe, hipStream_t s);
hipError_t launch_diag_chunk32(const LaunchCfg &cfg, const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F,
                               igdsp_frame_stats *stats, uint64_t *diag, hipStream_t s);
hipError_t launch_build_enc_table(int variant, uint8_t *tab, hipStream_t s);   // tab[law << 16 | uint16(v)] = enc(v), 131 072 bytes
hipError_t launch_encode(const LaunchCfg &cfg, const int16_t *pcm, const uint8_t *codec,
                         uint32_t C, uint32_t F, uint32_t n, uint8_t *out, int variant, hipStream_t s);
hipError_t launch_encode_table(const LaunchCfg &cfg, const int16_t *pcm, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                               uint8_t *out, int variant, hipStream_t s);
hipError_t launch_roundtrip(const LaunchCfg &cfg, int kernel_variant, const uint8_t *payload, const uint8_t *codec,
                            uint32_t C, uint32_t F, uint32_t n, uint8_t *out, igdsp_frame_stats *stats,
                            igdsp_chan_hold *hold, const uint8_t *gate, int variant, hipStream_t s);
hipError_t launch_hold_update(const igdsp_frame_stats *stats, const uint16_t *len, uint32_t C, uint32_t F, uint32_t n,
                              igdsp_chan_hold *hold, const uint8_t *gate, hipStream_t s);
hipError_t launch_flush_fold(const igdsp_frame_stats *stA, const igdsp_frame_stats *stB, const uint16_t *lenB, const uint2 *seq, const uint2 *runs,
                             uint32_t n_channels, uint32_t gate_mode, uint32_t alarm, igdsp_chan_hold *hold, igdsp_chan_probe *probe,
                             igdsp_frame_stats *last, hipStream_t s);
hipError_t launch_hold_reset(igdsp_chan_hold *hold, uint32_t C, const uint8_t *mask, hipStream_t s);
hipError_t launch_depayload(const LaunchCfg &cfg, const uint8_t *packets, const uint16_t *sizes, const uint8_t *radio,
                            uint32_t C, uint32_t F, uint32_t stride, uint32_t n, uint8_t *payload, uint16_t *len,
                            igdsp_rtp_info *info, hipStream_t s);
hipError_t launch_wav_expand(const LaunchCfg &cfg, const uint8_t *payload, uint32_t C, uint32_t F, uint32_t n, uint32_t rate,
                             uint8_t *files, uint64_t file_stride, hipStream_t s);
hipError_t launch_g726(const LaunchCfg &cfg, const uint8_t *in, uint8_t *out, uint64_t n_bytes, int mode, hipStream_t s);
hipError_t launch_gen_uniform(uint8_t *out, uint64_t n_bytes, uint64_t seed, uint64_t first_byte, hipStream_t s);
hipError_t launch_stream_rw(const LaunchCfg &cfg, const void *src, size_t bytes, void *dst, hipStream_t s);
hipError_t launch_stream_cluster(const LaunchCfg &cfg, const void *src, size_t bytes, void *dst, int k, hipStream_t s);
hipError_t launch_stream_pieces(const LaunchCfg &cfg, const void *src, uint32_t n_items, uint32_t stride, uint32_t hdr, int mode, int rows, void *dst, void *dst2, hipStream_t s);
hipError_t launch_stream_walk(const LaunchCfg &cfg, const void *src, uint32_t n_items, uint32_t stride, uint32_t hdr, uint32_t groups, uint32_t n_seg,
                              uint32_t trickle, void *dst, void *dst2, hipStream_t s);
hipError_t launch_stream_mix(const LaunchCfg &cfg, const void *src, void *dst, uint32_t n_items, int r, int w, int waves, hipStream_t s, void *dst2 = nullptr, const void *src2 = nullptr);
hipError_t launch_stream_read(const LaunchCfg &cfg, const void *src, size_t bytes, uint64_t *sink, hipStream_t s);

}  // namespace igdsp
