/*
 * igdsp.h — C ABI of the MI355X (gfx950) G.711 + level-meter hot path.
 *
 * This is the drop-in boundary for ONE path of piyanon108/iGate4xSoftphoneDSP:
 * the per-20 ms-RTP-frame work done behind the pjmedia transport adapter
 *   transport_rtp_cb  -> RoIP_ED137::setIncomingRTP(tp_adapter*)   (TransportAdapter.cpp:240-316, roip_ed137.cpp:6541-6587)
 *   transport_send_rtp -> RoIP_ED137::setOutgoingRTP(tp_adapter*)  (TransportAdapter.cpp:635-874, roip_ed137.cpp:6500-6536)
 * plus the G.711 decode/encode that pjmedia performs around those hooks
 * (TransportAdapter.cpp:301; codec selection roip_ed137.cpp:3546-3574) and the
 * level-meter contract of audiometer.cpp:30-31 / Functions.cpp:2126-2230.
 *
 * Plain C types only; usable from C99 and C++11 (the reference builds with
 * CONFIG += c++11, iGate4xSoftphoneDSP.pro:2).  No torch / HIP types appear
 * here: device buffers are `void*`-compatible raw pointers, streams are an
 * opaque `void*` (a hipStream_t; NULL is HIP's legacy default stream).
 *
 * Error convention follows the reference (pj_status_t, PJ_SUCCESS == 0,
 * TransportAdapter.cpp:135-223): every entry returns int, 0 == success,
 * negative == IGDSP_E*.  Nothing here throws or aborts the host.
 */
#ifndef IGDSP_H
#define IGDSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: igdsp_io_alloc / igdsp_io_free, igdsp_wav_expand, staging ring (igdsp_level.dropped, IGDSP_STAGE_DEPTH); every round-1 entry is
 * unchanged in signature and meaning.
 * 3: additive again — the ED-137 gated window (igdsp_window, igdsp_decode_meter_window, igdsp_window_update, igdsp_chan_probe),
 * igdsp_set_ed137 / igdsp_set_gate_mode / igdsp_get_probe on the single-frame path, igdsp_flush_begin / igdsp_flush_end. */
#define IGDSP_ABI_VERSION 3

/* ---- error codes (0 == PJ_SUCCESS-style success) ------------------------- */
#define IGDSP_OK          0
#define IGDSP_EINVAL    (-22)  /* bad argument (NULL, size, alignment, unknown codec)   */
#define IGDSP_ENOMEM    (-12)  /* host or device allocation failed                      */
#define IGDSP_ENODEV    (-19)  /* no usable gfx950 device / HIP runtime error            */
#define IGDSP_ENOENT     (-2)  /* call_id not mapped to a channel (a4 routing miss)      */
#define IGDSP_ERANGE    (-34)  /* channel / frame index out of the context's capacity    */
#define IGDSP_EBUSY     (-16)  /* staging ring of the channel was full: the OLDEST staged frame was overwritten */
#define IGDSP_EDEVICE   (-5)   /* kernel launch / runtime failure (see igdsp_last_error) */

/* ---- codec ids: RTP payload types, as gated at TransportAdapter.cpp:252 ---
 * (WAVE_FORMAT_MULAW 0x0007 / WAVE_FORMAT_ALAW 0x0006 of Codecs.h:33-34 are the
 * container tags; on the wire and in this ABI the RTP PT is the codec id.) */
#define IGDSP_PT_PCMU      0   /* G.711 mu-law */
#define IGDSP_PT_PCMA      8   /* G.711 A-law  */
#define IGDSP_PT_R2S     123   /* ED-137 keep-alive: never metered (TransportAdapter.cpp:299,308) */

/* ---- frame geometry (roip_ed137.h:112,115: CLOCK_RATE 8000, PTIME 20) ----- */
#define IGDSP_CLOCK_RATE          8000
#define IGDSP_PTIME_MS              20
#define IGDSP_SAMPLES_PER_FRAME    160
#define IGDSP_MAX_PAYLOAD          256   /* tp_adapter::payload_buff[256], TransportAdapter.h:66 */
#define IGDSP_STAGE_DEPTH            8   /* frames a channel can stage between two igdsp_flush calls (160 ms of audio) */
#define IGDSP_METER_FULL_SCALE   30000   /* audiometer.cpp:30-31 */

/* ---- G.711 encoder variant -------------------------------------------------
 * Decode is unique (ITU-T G.711 tables).  Two historic encoders exist; they
 * agree on every value a decoder can produce and on ITU decision values, and
 * differ only in rounding of some negative inputs / clipping:
 *   SUN16 : 16-bit-domain Sun g711.c lineage (mu-law: BIAS 0x84 added to the
 *           16-bit magnitude, clip 32635; A-law: "-pcm - 8" with the < 0 clamp).
 *   G191  : 14/13-bit-domain ITU-T G.191 STL lineage (mu-law: pcm >> 2, BIAS 0x21,
 *           clip 8159; A-law: pcm >> 3, "-pcm - 1") — bit-identical to CPython
 *           `audioop`, which pins it exhaustively (tests/golden/g711_audioop.npz).
 * WHICH of the two the reference's pjmedia build carries is UNVERIFIED: pjmedia is
 * a third-party dependency of the reference, unpinned and absent from this tree
 * (SURVEY.md 8c), and the reference holds no G.711 vector.  A maintainer with
 * pjmedia's source picks the variant by reading pjmedia/src/pjmedia/alaw_ulaw.c
 * against the two descriptions above (a ">> 2" / ">> 3" before the segment search
 * = G191; a 16-bit BIAS 0x84 / "- 8" = SUN16).  Every binding and example of this
 * repository defaults to G191, the one an independent implementation pins; the
 * round trip of DECODED codes is identical in both (closed set, tested).
 */
#define IGDSP_ENC_SUN16   0
#define IGDSP_ENC_G191    1

/* ---- per-frame flags ------------------------------------------------------- */
#define IGDSP_FLAG_SILENT   0x01  /* peak <= 8: digital silence (A 0xD5/0x55, mu 0xFF/0x7F/0xFE/0x7E) */
#define IGDSP_FLAG_PROBE_D5 0x02  /* payload[28]==payload[38]==payload[48]==0xD5: the reference's own
                                     TX silence probe on packet bytes 40/50/60 behind a 12-byte RTP
                                     header (TransportAdapter.cpp:657-673) */
#define IGDSP_FLAG_CLIPPED  0x04  /* peak == codec full scale (32124 mu / 32256 A) */
#define IGDSP_FLAG_EMPTY    0x08  /* zero-length frame (missing / keep-alive slot): all fields 0 */

/* Per channel-frame result, 16 bytes, naturally aligned.
 *  sumsq     : exact sum of x^2 over the decoded int16 samples of the frame
 *  rms       : sqrtf((float)sumsq / n)                 (fp32; oracle is float64, rel tol 1e-5)
 *  peak      : max |x|                                  (<= 32256)
 *  byte_mean : (uint8_t)(sum of raw payload bytes / n)  — bit-exact restatement of the
 *              reference's "audioLevel" (roip_ed137.cpp:6564-6568, unsigned-char target)
 *  flags     : IGDSP_FLAG_*                                                          */
typedef struct igdsp_frame_stats {
    uint64_t sumsq;
    float    rms;
    uint16_t peak;
    uint8_t  byte_mean;
    uint8_t  flags;
} igdsp_frame_stats;

/* Per-channel running aggregate / peak-hold, 32 bytes.  Mirrors the PTT-window
 * logger (keeplogAudioLevel Functions.cpp:2126-2145; reset createPTTEventDataLogger
 * Functions.cpp:2155-2167): while a channel's window is open each frame does
 * count++, level_sum += byte_mean, level_max/min update, plus (new) sumsq_acc and
 * peak_hold for the decoded-domain meter.  Reset state: all 0, level_min = 255. */
typedef struct igdsp_chan_hold {
    uint64_t sumsq_acc;   /* sum of frame sumsq over the window                       */
    uint32_t count;       /* frames accumulated (trx::level_in_count)                 */
    uint32_t level_sum;   /* exact sum of byte_mean; reference's uint16 OutgoingRTPSum
                             (roip_ed137.h:741) equals (uint16_t)level_sum             */
    uint32_t samples;     /* samples accumulated (count * n for full frames)          */
    uint16_t peak_hold;   /* running max |x|                                          */
    uint8_t  level_max;   /* trx::OutgoingRTPmax (init 0)                             */
    uint8_t  level_min;   /* trx::OutgoingRTPmin (init 255)                           */
    uint32_t n_silent;    /* frames flagged IGDSP_FLAG_SILENT                         */
    uint32_t n_clipped;   /* frames flagged IGDSP_FLAG_CLIPPED                        */
} igdsp_chan_hold;

/* Node/launch aggregate: the packed vector of SURVEY 8(e).  All u64 so that ONE
 * sum all-reduce (RCCL ncclSum over int64) yields sums AND the max: rank g writes
 * its local peak only into peak_slot[g]; after the sum every rank holds all peaks. */
#define IGDSP_AGG_MAX_RANKS 8
/* Every counter owns a 128-byte line: a persistent launch ends with one device atomic per block per counter, and
 * atomics on ONE line serialise (~95 per microsecond) - seven counters sharing a line cost ~8 us per launch on
 * MI355X, on separate lines ~3 us (tools/agg_ab.py).  The padding is zero, so the struct is still summed as one
 * vector of IGDSP_AGG_WORDS uint64. */
#define IGDSP_AGG_LINE_WORDS 16
typedef struct igdsp_aggregate {
    uint64_t sumsq;         uint64_t pad0[IGDSP_AGG_LINE_WORDS - 1];   /* sum of sumsq over all frames      */
    uint64_t samples;       uint64_t pad1[IGDSP_AGG_LINE_WORDS - 1];   /* samples metered                   */
    uint64_t frames;        uint64_t pad2[IGDSP_AGG_LINE_WORDS - 1];   /* non-empty frames metered          */
    uint64_t n_silent;      uint64_t pad3[IGDSP_AGG_LINE_WORDS - 1];
    uint64_t n_clipped;     uint64_t pad4[IGDSP_AGG_LINE_WORDS - 1];
    uint64_t byte_mean_sum; uint64_t pad5[IGDSP_AGG_LINE_WORDS - 1];   /* sum of byte_mean (checksum-of-checksums) */
    uint64_t peak_slot[IGDSP_AGG_MAX_RANKS];                            /* this rank's peak in slot[rank]    */
    uint64_t pad6[IGDSP_AGG_LINE_WORDS - IGDSP_AGG_MAX_RANKS];
} igdsp_aggregate;
#define IGDSP_AGG_WORDS (7 * IGDSP_AGG_LINE_WORDS)

/* What igdsp_poll returns for one channel: everything host code needs to fill
 * trx::IncomingRTP (roip_ed137.cpp:6570-6585) and feed updateInputLevel(int percent)
 * (roip_ed137.cpp:584-592; scale audiometer.cpp:30-31). */
typedef struct igdsp_level {
    uint8_t  byte_mean;   /* -> trx->radioN->IncomingRTP / OutgoingRTP               */
    uint8_t  flags;
    uint16_t peak;
    float    rms;
    int32_t  percent;     /* int(float(rms*100.0/30000.0)), AudioMeter::onValueChanged */
    uint16_t peak_hold;
    uint16_t dropped;     /* frames overwritten in the staging ring before a flush took them (saturates at 65535) */
    uint32_t frames;      /* frames metered for this channel since create            */
} igdsp_level;

typedef struct igdsp_ctx igdsp_ctx;

/* ---- lifecycle -------------------------------------------------------------- */

/* Create a context on HIP device `device` with staging capacity for
 * `max_channels` concurrent calls.  Allocates the pinned host slab
 * [max_channels][160], its device mirror, result buffers and a private stream.
 * Fails with IGDSP_ENODEV when no gfx950 device/runtime is usable — there is
 * no CPU fallback. */
int igdsp_create(igdsp_ctx **out, int device, uint32_t max_channels);
int igdsp_destroy(igdsp_ctx *ctx);
/* Human-readable text of the last runtime error on this context ("" if none). */
const char *igdsp_last_error(const igdsp_ctx *ctx);
int igdsp_abi_version(void);
/* Device the context is bound to; number of CUs (for grid sizing reports). */
int igdsp_device_info(const igdsp_ctx *ctx, int *device, int *compute_units, char *name, size_t name_len);

/* ---- a4: call-id -> channel routing (roip_ed137.cpp:6519-6534, 6570-6585) ---- */
int igdsp_map_call(igdsp_ctx *ctx, int32_t call_id, uint32_t channel);
int igdsp_unmap_call(igdsp_ctx *ctx, int32_t call_id);

/* ---- (i) single-frame entry, callable from setIncomingRTP/setOutgoingRTP -------
 * Inputs are exactly what those hooks read from tp_adapter: callID, payload
 * pointer, payload length (roip_ed137.cpp:6549-6552) and the RTP PT
 * (TransportAdapter.cpp:252).  Copies `payload` (borrowed; pjmedia owns pkt) into
 * the channel's staging ring (IGDSP_STAGE_DEPTH frames deep: the reference's hook runs on every frame,
 * TransportAdapter.cpp:303, and so every frame reaches the meter) and returns; never blocks on the device.  Safe to
 * call concurrently from several media threads for DIFFERENT channels.  If the owner thread has not flushed for more
 * than IGDSP_STAGE_DEPTH frames the oldest staged frame of the channel is overwritten, counted in igdsp_level.dropped,
 * and the call returns IGDSP_EBUSY (the new frame IS staged).  pt == 123 (R2S keep-alive) and unknown PTs are accepted
 * and ignored (returns 0, nothing staged), like the reference which meters only pt != 123.  payloadlen > 256 ->
 * IGDSP_EINVAL (the reference would overflow payload_buff[256] there, TransportAdapter.cpp:286). */
int igdsp_on_rtp_frame(igdsp_ctx *ctx, int32_t call_id, uint8_t pt,
                       const uint8_t *payload, uint32_t payloadlen);

/* Take every frame staged since the previous flush (all of them, oldest first per channel), upload them compacted in
 * one copy, meter them — whole 160-byte frames through the tuned chunk kernel once 64 or more are staged, everything else
 * through the general kernel — fold EVERY frame into the per-channel hold state (keeplogAudioLevel semantics per frame,
 * Functions.cpp:2126-2145) and make each channel's newest record visible to igdsp_poll.  Called by ONE owner thread per
 * context (e.g. the reference's 40 ms timer, roip_ed137.cpp:1756).  `n_frames_out` (optional) receives the number of
 * staged frames processed. */
int igdsp_flush(igdsp_ctx *ctx, uint32_t *n_frames_out);

/* Non-blocking form of igdsp_flush, for owner threads that must not wait (the reference's hook and its 40 ms timer slot never
 * wait: TransportAdapter.cpp:303, roip_ed137.cpp:1756).  igdsp_flush_begin snapshots the staged frames, enqueues upload, kernels
 * and downloads on the context's stream and returns; igdsp_flush_end waits for that work (normally finished long before the next
 * tick) and publishes the levels for igdsp_poll.  wait == 0: return IGDSP_EBUSY instead of waiting if the device has not finished.
 * igdsp_flush == begin + end(wait = 1).  A begin while the previous flush is still open ends it first (waiting).  igdsp_poll /
 * igdsp_get_hold / igdsp_get_probe read the PUBLISHED snapshot and never wait for a flush in flight. */
int igdsp_flush_begin(igdsp_ctx *ctx, uint32_t *n_frames_out);
int igdsp_flush_end(igdsp_ctx *ctx, int wait);

/* The third hook of the reference's boundary, setIncomingED137Value(uint32_t value, pjsua_acc_id) (roip_ed137.h:273; called by
 * transport_rtp_cb with ntohl(adapter->ed137_value), TransportAdapter.cpp:305,313): records the call's current ED-137 word (host
 * order).  Frames staged by igdsp_on_rtp_frame AFTER it carry that word, and igdsp_flush folds a frame into the call's window only
 * if the word passes the context's gate mode (igdsp_set_gate_mode: IGDSP_GATE_* below; default IGDSP_GATE_ALWAYS = every frame,
 * the round-2 behaviour).  Wait-free for the caller, like igdsp_on_rtp_frame.  igdsp_get_probe: the call's consecutive-silence
 * state (adapter->rtpFalse, TransportAdapter.cpp:657-673) as of the last finished flush. */
int igdsp_set_ed137(igdsp_ctx *ctx, int32_t call_id, uint32_t ed137_value);
int igdsp_set_gate_mode(igdsp_ctx *ctx, uint32_t gate_mode);
struct igdsp_chan_probe;
int igdsp_get_probe(igdsp_ctx *ctx, uint32_t channel, struct igdsp_chan_probe *out);

/* (iii) results poll for one channel (valid after a flush). */
int igdsp_poll(igdsp_ctx *ctx, uint32_t channel, igdsp_level *out);
int igdsp_poll_call(igdsp_ctx *ctx, int32_t call_id, igdsp_level *out);
/* (iv) reset the peak-hold / window aggregate of one channel (PTT press,
 * Functions.cpp:2155-2167). channel == UINT32_MAX resets all. */
int igdsp_reset_hold(igdsp_ctx *ctx, uint32_t channel);
/* Read back the hold state of one channel. */
int igdsp_get_hold(igdsp_ctx *ctx, uint32_t channel, igdsp_chan_hold *out);

/* ---- (ii) batched device entries ----------------------------------------------
 * All d_* pointers are DEVICE pointers on the context's device.  Layout is
 * time-major, as frames arrive: payload[f][c][n] u8, codec[c] u8 (RTP PT 0 / 8),
 * stats[f][c], pcm[f][c][n] i16.  n = samples_per_frame (1..256; 160 is the tuned
 * path).  d_len (optional, may be NULL) gives a per-frame valid length
 * len[f][c] <= n for ragged input; bytes past len are ignored; len 0 marks an
 * empty slot.  Work is enqueued on `stream` (a hipStream_t; NULL = the legacy
 * default stream) and NOT synchronised. */

/* a1+a3+a5+a7: decode + meter.  d_pcm may be NULL (meter-only, the headline).
 * d_agg (optional) is an igdsp_aggregate on the device that this launch ADDS
 * into (zero it first with igdsp_agg_reset); rank selects the peak slot. */
int igdsp_decode_meter(igdsp_ctx *ctx,
                       const uint8_t *d_payload, const uint8_t *d_codec, const uint16_t *d_len,
                       uint32_t n_channels, uint32_t n_frames, uint32_t samples_per_frame,
                       igdsp_frame_stats *d_stats, int16_t *d_pcm,
                       igdsp_aggregate *d_agg, uint32_t rank, void *stream);

/* a2: encode int16 PCM -> G.711 codes, per-channel law. variant = IGDSP_ENC_*. */
int igdsp_encode(igdsp_ctx *ctx,
                 const int16_t *d_pcm, const uint8_t *d_codec,
                 uint32_t n_channels, uint32_t n_frames, uint32_t samples_per_frame,
                 uint8_t *d_payload_out, int variant, void *stream);

/* a1+a2+a5+a6 fused (config #5): decode -> stats -> re-encode -> per-channel hold.
 * d_gate[c] (optional): 0 = window closed (frame metered but not folded into
 * hold), non-zero = open.  d_hold[c] persists across launches. */
int igdsp_roundtrip_peakhold(igdsp_ctx *ctx,
                             const uint8_t *d_payload, const uint8_t *d_codec,
                             uint32_t n_channels, uint32_t n_frames, uint32_t samples_per_frame,
                             uint8_t *d_payload_out, igdsp_frame_stats *d_stats,
                             igdsp_chan_hold *d_hold, const uint8_t *d_gate,
                             int variant, void *stream);

/* Fold stats[f][c] into hold[c] (for callers that ran igdsp_decode_meter). */
int igdsp_hold_update(igdsp_ctx *ctx, const igdsp_frame_stats *d_stats,
                      uint32_t n_channels, uint32_t n_frames, uint32_t samples_per_frame,
                      igdsp_chan_hold *d_hold, const uint8_t *d_gate, void *stream);
int igdsp_hold_reset(igdsp_ctx *ctx, igdsp_chan_hold *d_hold, uint32_t n_channels,
                     const uint8_t *d_reset_mask, void *stream);

int igdsp_agg_reset(igdsp_ctx *ctx, igdsp_aggregate *d_agg, void *stream);

/* ---- ED-137 RTP depayload + gather on the device ------------------
 * The step BEFORE the path: transport_rtp_cb's header parse and payload copy
 * (TransportAdapter.cpp:240-292; header layout ed137_rtp.h:22-48), batched.
 * Input  : packets[f][c][pkt_stride] raw RTP packets as received (slot of pkt_stride bytes,
 *          pkt_stride % 4 == 0), sizes[f][c] = received size (NULL: every packet fills its slot),
 *          radio[c] != 0 => 20-byte ED-137 header (12 B RTP + ext hdr 0x0167/len 1 + ED-137 word),
 *          else plain 12-byte RTP (TransportAdapter.cpp:270,279).
 * Output : payload[f][c][n] dense (bytes past the payload length zeroed), len[f][c] = metered payload
 *          length (0 for PT 123 keep-alives, non-G.711 PTs, runts, and oversize packets — the
 *          reference drops payloads it cannot buffer, TransportAdapter.cpp:286-291), info[f][c].
 * `len` feeds igdsp_decode_meter's d_len directly. */
typedef struct igdsp_rtp_info {
    uint32_t ed137;        /* ntohl(ED-137 word), 0 on non-radio calls (get_ed137_value, TransportAdapter.cpp:337-346) */
    uint16_t payload_len;  /* size - header, before any clamp (what transport_rtp_cb stores in payload_bufSize)       */
    uint8_t  pt;           /* RTP payload type (7 bits)                                                              */
    uint8_t  flags;        /* IGDSP_RTP_*                                                                            */
} igdsp_rtp_info;
#define IGDSP_RTP_V2        0x01  /* version field == 2                                   */
#define IGDSP_RTP_X         0x02  /* header-extension bit                                 */
#define IGDSP_RTP_MARKER    0x04
#define IGDSP_RTP_ED137_OK  0x08  /* radio call, X set, profile 0x0167, length 1          */
#define IGDSP_RTP_KEEPALIVE 0x10  /* PT 123 (R2S): never metered                          */
#define IGDSP_RTP_METERED   0x20  /* PT 0 or 8 with a usable payload: len[f][c] > 0       */
#define IGDSP_RTP_RUNT      0x40  /* size < header                                        */
#define IGDSP_RTP_OVERSIZE  0x80  /* payload longer than n: dropped                       */
/* ED-137 word fields, masks as the reference extracts them (Functions.cpp:1018,1045,1136,1148) */
#define IGDSP_ED137_PTT_TYPE(v) (((v) & 0xe0000000u) >> 29)
#define IGDSP_ED137_SQU(v)      (((v) & 0x10000000u) >> 28)
#define IGDSP_ED137_PTT_ID(v)   (((v) & 0x0fc00000u) >> 22)
#define IGDSP_ED137_BSS(v)      (((v) & 0x000000f8u) >> 3)

int igdsp_depayload(igdsp_ctx *ctx, const uint8_t *d_packets, const uint16_t *d_sizes, const uint8_t *d_radio,
                    uint32_t n_channels, uint32_t n_frames, uint32_t pkt_stride, uint32_t samples_per_frame,
                    uint8_t *d_payload_out, uint16_t *d_len_out, igdsp_rtp_info *d_info_out, void *stream);

/* ---- Fused fast path: packet slots straight into the meter (depayload + decode + meter in ONE kernel) ----
 * For receivers that deposit radio-call packets (20-byte ED-137 header + 160-byte G.711 payload = 180 bytes,
 * TransportAdapter.cpp:846) into fixed 192-byte slots laid out so the payload is 16-byte aligned:
 *     bytes 0..1   received packet size (uint16 LE)         bytes 2..11  reserved (0)
 *     bytes 12..31 the 20-byte header (custom_rtp_hdr)      bytes 32..191 payload
 * slots[f][c][192].  A frame is metered iff size == 180 and its RTP PT equals codec[c] (0 or 8); every
 * other slot (PT 123 keep-alive, other PT, other size) gets an IGDSP_FLAG_EMPTY record — route those
 * rare frames through igdsp_depayload + igdsp_decode_meter(d_len) if they must be metered.
 * d_info (optional) receives the same igdsp_rtp_info igdsp_depayload would produce. */
#define IGDSP_SLOT_BYTES      192
#define IGDSP_SLOT_HDR_OFFSET  12
#define IGDSP_SLOT_PAYLOAD_OFFSET 32
int igdsp_decode_meter_rtp(igdsp_ctx *ctx, const uint8_t *d_slots, const uint8_t *d_codec,
                           uint32_t n_channels, uint32_t n_frames,
                           igdsp_frame_stats *d_stats, igdsp_rtp_info *d_info,
                           igdsp_aggregate *d_agg, uint32_t rank, void *stream);

/* The same fused kernel over packets packed exactly as igdsp_depayload takes them: packets[f][c][pkt_stride]
 * (pkt_stride % 4 == 0, >= hdr_bytes + 160), sizes[f][c] (NULL: every packet is hdr_bytes + 160 long), ONE header
 * type per launch (hdr_bytes = 20 for ED-137 radio legs, 12 for plain SIP legs).  Piece addresses are only dword
 * aligned here; gfx950 global loads need no more.  Metered iff size == hdr_bytes + 160 and PT == codec[c]. */
int igdsp_decode_meter_packets(igdsp_ctx *ctx, const uint8_t *d_packets, const uint16_t *d_sizes, const uint8_t *d_codec,
                               uint32_t n_channels, uint32_t n_frames, uint32_t pkt_stride, uint32_t hdr_bytes,
                               igdsp_frame_stats *d_stats, igdsp_rtp_info *d_info,
                               igdsp_aggregate *d_agg, uint32_t rank, void *stream);

/* The same again with a PER-CHANNEL header length: d_radio[c] != 0 -> 20-byte ED-137 header, 0 -> 12-byte RTP header
 * (radio legs and plain SIP legs in one launch, as in the reference's process, TransportAdapter.cpp:265-292).
 * pkt_stride >= 180.  Metered iff size == header + 160 and PT == codec[c]; info as igdsp_depayload would give it. */
int igdsp_decode_meter_packets_mixed(igdsp_ctx *ctx, const uint8_t *d_packets, const uint16_t *d_sizes, const uint8_t *d_codec,
                                     const uint8_t *d_radio, uint32_t n_channels, uint32_t n_frames, uint32_t pkt_stride,
                                     igdsp_frame_stats *d_stats, igdsp_rtp_info *d_info,
                                     igdsp_aggregate *d_agg, uint32_t rank, void *stream);

/* ---- SURVEY 8(f) rank 1, last clause: squelch / PTT gating of the meter from the ED-137 word, on the device ---------------------
 * The reference folds a frame's level into the PTT window (keeplogAudioLevel, Functions.cpp:2126-2145) while the window is open,
 * and reads PTT / squelch out of the ED-137 word of the RTP header extension (masks: PTT type Functions.cpp:1136, PTT id :1148,
 * SQU :1160, BSS :1018).  A window here is the per-channel igdsp_chan_hold; a metered frame (f, c) is folded into hold[c] iff
 *     (d_gate == NULL || d_gate[c] != 0)                     the caller's per-channel window state (eventPttSQL_In_LoggingOn)
 *  && frame_gate(gate_mode, ED-137 word of the frame's OWN packet)
 * with frame_gate = 1 (ALWAYS), SQU bit set (SQU), PTT type != 0 (PTT), either (SQU_OR_PTT).  Legs without an ED-137 word
 * (12-byte RTP header: the word reads 0, as get_ed137_value gives it) are therefore never folded under SQU / PTT gating.
 * Independently of the gate, d_probe[c] follows the reference's consecutive-silence counter (adapter->rtpFalse,
 * TransportAdapter.cpp:657-673): a metered frame long enough to hold the probe bytes (payload length > 48) adds 1 to `run` when
 * its IGDSP_FLAG_PROBE_D5 is set and clears `run` when it is not; shorter / empty frames leave it; `alarms` counts how often
 * `run` reached probe_alarm (the reference logs at exactly 500).  Frames of a channel are taken in frame order f = 0 .. F-1. */
#define IGDSP_GATE_ALWAYS      0u
#define IGDSP_GATE_SQU         1u
#define IGDSP_GATE_PTT         2u
#define IGDSP_GATE_SQU_OR_PTT  3u
#define IGDSP_PROBE_ALARM    500u   /* TransportAdapter.cpp:666 */
typedef struct igdsp_chan_probe {
    uint32_t run;        /* consecutive probe-matching frames up to now (adapter->rtpFalse)   */
    uint32_t alarms;     /* times run reached the alarm length                                 */
} igdsp_chan_probe;
typedef struct igdsp_window {
    uint32_t gate_mode;           /* IGDSP_GATE_*                                                                    */
    uint32_t probe_alarm;         /* 0 = IGDSP_PROBE_ALARM                                                           */
    igdsp_chan_hold *d_hold;      /* [C] window aggregate the gated frames are folded into (required)                */
    const uint8_t *d_gate;        /* [C] optional per-channel window state, 0 = closed                               */
    igdsp_chan_probe *d_probe;    /* [C] optional consecutive-silence state                                          */
    void *d_work;                 /* igdsp_window_work_bytes(C) bytes of device scratch, 16-byte aligned (48 B x 8 x C): */
                                  /* igdsp_decode_meter_window's fused kernels need it (the per-segment window / run     */
                                  /* summaries of the form that keeps the windows in registers); not used by             */
                                  /* igdsp_window_update.  One buffer per stream that launches                           */
} igdsp_window;
size_t igdsp_window_work_bytes(uint32_t n_channels);

/* Fold records into the window: the generalisation of igdsp_hold_update to per-FRAME gates.  d_info (optional) = the
 * igdsp_rtp_info[F][C] of igdsp_depayload / the fused packet entries: supplies each frame's ED-137 word (NULL: word 0) and,
 * when d_len is NULL, its length (payload_len clamped to samples_per_frame; NULL as well: samples_per_frame).  EMPTY records
 * are skipped.  One thread per channel walks its F frames in order. */
int igdsp_window_update(igdsp_ctx *ctx, const igdsp_frame_stats *d_stats, const igdsp_rtp_info *d_info, const uint16_t *d_len,
                        uint32_t n_channels, uint32_t n_frames, uint32_t samples_per_frame, const igdsp_window *win, void *stream);

/* The fused packet entries with the window folded in the SAME launch: packets -> records (+ info, aggregate) -> hold[c] and
 * probe[c], one pass over the packets.  layout selects the packet format and which of the arguments apply:
 *   IGDSP_PKT_SLOTS  = igdsp_decode_meter_rtp            (192-byte slots; d_sizes, d_radio, pkt_stride, hdr_bytes ignored)
 *   IGDSP_PKT_PACKED = igdsp_decode_meter_packets        (pkt_stride, hdr_bytes 12 / 20, optional d_sizes; d_radio ignored)
 *   IGDSP_PKT_MIXED  = igdsp_decode_meter_packets_mixed  (pkt_stride, d_radio, optional d_sizes; hdr_bytes ignored)
 * Same argument rules and records as those entries.  With n_channels % 64 == 0 and win->d_work given the windows are kept on
 * the chip during the launch, in one of two forms the library picks by shape: a workgroup owns 64 / 128 / 256 consecutive
 * channels for the launch, keeps their windows and silence runs in its LDS and writes hold[c] / probe[c] itself (launches of
 * more than 255 frames go out in parts on the stream); or a wavefront keeps 64 channels' windows and runs in registers over a
 * segment of the frames, the segments' summaries go through d_work and are folded in frame order by a small second kernel.
 * Integer sums / max / min and an in-order run either way: bit-identical to the sequential fold.  Other channel counts, or
 * d_work == NULL, run the plain fused kernel followed by igdsp_window_update on the same stream; d_info has to be given then
 * (it carries each frame's ED-137 word and length).  On the fused path d_stats may be NULL: a host that only wants the windows
 * (the PTT logger) then pays for no per-frame record at all — the launch reads the packets and writes hold / probe. */
#define IGDSP_PKT_SLOTS   0u
#define IGDSP_PKT_PACKED  1u
#define IGDSP_PKT_MIXED   2u
int igdsp_decode_meter_window(igdsp_ctx *ctx, uint32_t layout, const uint8_t *d_packets, const uint16_t *d_sizes,
                              const uint8_t *d_codec, const uint8_t *d_radio, uint32_t n_channels, uint32_t n_frames,
                              uint32_t pkt_stride, uint32_t hdr_bytes, igdsp_frame_stats *d_stats, igdsp_rtp_info *d_info,
                              igdsp_aggregate *d_agg, uint32_t rank, const igdsp_window *win, void *stream);

/* ---- recorder-compatible output on the device (WavWriter.cpp:63-156) ------------------------------
 * The step AFTER the path.  For every channel c of a batch payload[F][C][n] one complete file image exactly as the
 * reference's recorder would leave it after writeRTPWav(frame 0) ... writeRTPWav(frame F-1), stop():
 *     files + c * file_stride : [ 44-byte header | every payload byte b as the two bytes {b, 0x00} ]   (44 + 2 F n bytes)
 * header = WavWriter::start's (RIFF / WAVE / "fmt " 16, tag 7, "channels" 2, rate, rate * 4, align 4, 16 bits, "data") with
 * the two sizes WavWriter::stop patches in (36 + 2 F n and 2 F n).  file_stride >= 44 + 2 F n; a multiple of 4 (and n % 8
 * == 0) takes the tiled streaming kernel, anything else a byte-wise one.  Fastest (0.77 of the HBM peak instead of 0.70)
 * when (d_files + 44) % 128 == 0 and file_stride % 128 == 0: the payload-derived bytes of every file are then line-aligned.  Byte-identical to the host recorder
 * (igdsp_wav_* in the host mirror) and to the REAL WavWriter.cpp (tests/golden/config1_4ch_50f.npz). */
int igdsp_wav_expand(igdsp_ctx *ctx, const uint8_t *d_payload, uint32_t n_channels, uint32_t n_frames, uint32_t samples_per_frame,
                     uint32_t rate, uint8_t *d_files, uint64_t file_stride, void *stream);

/* ---- G.726 code-word reorder (RoIP_ED137::changeUplinkOrder, roip_ed137.cpp:6379-6499) ----
 * Repacks G.726 code words between the RFC 3551 and AAL2 bit orders, bug-for-bug as the reference
 * does it on its (unsigned-char) target:
 *   mode 1 (16 kbit/s, 2-bit): the four 2-bit fields of every byte are reversed          (:6382-6389)
 *   mode 2 (24 kbit/s, 3-bit): 3-byte groups through the reference's bit-field struct   (:6390-6441)
 *   mode 3 (32 kbit/s, 4-bit): nibbles swapped                                           (:6442-6449)
 *   mode 4 (40 kbit/s, 5-bit): 5-byte groups; the reference's 2-bit field S2_ receives
 *          `(b0 >> 5) & 0x04`, which never fits, so those two output bits are always 0  (:6450-6498)
 * n_bytes must be a multiple of the group size (1, 3, 1, 5): the reference over-reads otherwise. */
int igdsp_g726_reorder(igdsp_ctx *ctx, const uint8_t *d_in, uint8_t *d_out, uint64_t n_bytes, int g726UplinkBitrate, void *stream);

/* ---- synthetic input generators (device side; SURVEY 8(d) definitions) ---------
 * D-uniform: byte k of global byte index g is
 *   (splitmix64(seed + (g>>3)) >> (8*(g&7))) & 0xFF,  g = first_byte + k
 * so any shard of the [F][C][n] array is reproducible on any GPU count. */
int igdsp_gen_uniform(igdsp_ctx *ctx, uint8_t *d_out, uint64_t n_bytes,
                      uint64_t seed, uint64_t first_byte, void *stream);

/* ---- small device-memory helpers for hosts that do not carry a HIP runtime
 * of their own (the Qt/C++ softphone).  Thin wrappers; all synchronous except
 * where a stream is given. */
int igdsp_dev_alloc(igdsp_ctx *ctx, void **d_ptr, size_t bytes);
int igdsp_dev_free(igdsp_ctx *ctx, void *d_ptr);
/* (Round-1 helper, superseded by igdsp_io_alloc below which places a whole buffer set and also finds the third class.)
 * Allocate an OUTPUT buffer in another class of device memory than the input it will be written from (see
 * igdsp_probe_placement): tries up to max_tries positions, each a further spacer_bytes (0 = 12 GiB) of temporary
 * allocation away, times the bare read(d_in) + write(candidate) stream for each, keeps the fastest candidate, frees the
 * rest and the spacers.  Stops early once a candidate is >= 8 % faster than the first.  ms_first / ms_kept (optional)
 * return the probe times of the plain first allocation and of the one kept.  Synchronous; start-up use only. */
int igdsp_dev_alloc_far(igdsp_ctx *ctx, void **d_ptr, size_t bytes, const void *d_in, size_t in_bytes,
                        uint32_t max_tries, size_t spacer_bytes, float *ms_first, float *ms_kept);
/* ---- placement-aware allocation of a whole input / output buffer set ---------------------------------------------------
 * On MI355X a launch that reads one class of device memory and writes another is ~13 % faster than one that reads and
 * writes the same class, and a bulk write stream spread over the two classes the inputs are NOT in gains another 5-8 %
 * (igdsp_probe_placement below).  Which class an allocation lands in cannot be queried and differs per
 * process; consecutive plain allocations normally share one.  igdsp_io_alloc therefore takes the whole buffer set of a
 * pipeline stage at once, classifies 128 MiB chunks of physical device memory by timing the bare read + record-store
 * stream against the INPUT buffers, and maps chunks of the right class behind each buffer's (contiguous) address range:
 *   IGDSP_IO_INPUT   buffers the kernels read (payload ring, packet ring, PCM to encode): consecutive chunks, class "A";
 *   IGDSP_IO_RECORD  small written outputs (igdsp_frame_stats, igdsp_rtp_info, len, hold): a class other than A;
 *   IGDSP_IO_BULK    large written outputs (PCM, re-encoded / dense payload): first half in one non-A class, second half
 *                    in the other (the kernels visit the two halves of a bulk output alternately).
 * Start-up use: synchronous, the first call takes 0.2-10 s (the search, then report->settle_ms of waiting until the driver has finished
 * clearing the memory the search gave back: launches run 1-5 % slow while it does); temporarily holds up to
 * explore_limit_bytes (0 = 50 % of the free device memory, 85 % when BULK buffers want a third class) of chunks while it searches and releases everything it does not
 * hand out — except up to 16 spare chunks (2 GiB) per memory class, which stay with the context (as do the chunks of a set given back with igdsp_io_free): a later
 * call that the spares cover is placed in < 100 ms without probing or waiting; igdsp_destroy gives them back (IGDSP_IO_SPARE_CHUNKS=0: keep none).  The pointers stay valid until igdsp_io_free.  If the virtual-memory API is missing, the largest input is < 512 MiB (the probe then measures the
 * Infinity Cache, and placement does not matter) or no second class is found, the buffers are still allocated and
 * report->placed is 0.  Buffer sizes are rounded up to whole chunks internally. */
#define IGDSP_IO_INPUT   0u
#define IGDSP_IO_RECORD  1u
#define IGDSP_IO_BULK    2u
typedef struct igdsp_io_buf {
    size_t   bytes;      /* in  */
    uint32_t role;       /* in: IGDSP_IO_* */
    uint32_t reserved;
    void    *ptr;        /* out: device pointer (chunk aligned) */
} igdsp_io_buf;
typedef struct igdsp_io_report {
    uint32_t placed;           /* 1: every RECORD / BULK buffer sits in another class than the inputs                  */
    uint32_t bulk_spread;      /* 1: BULK buffers have their halves in two different non-input classes                 */
    uint32_t classes_found;    /* 0 (no probing done), 1, 2 or 3                                                       */
    uint32_t chunks_explored;  /* chunks created while searching (most are released again)                            */
    uint32_t probes;           /* timed probe rounds                                                                   */
    uint32_t reseeds;          /* times the probe source was re-seeded because it straddled a class boundary          */
    uint64_t chunk_bytes;
    uint64_t explored_bytes;
    float    probe_ms_same;    /* bare probe stream (1.25 GiB read + 1/10 written) writing into the class it reads     */
    float    probe_ms_other;   /* ... writing into another class                                                       */
    float    setup_ms;         /* host wall time of the call                                                           */
    float    settle_ms;        /* of setup_ms: waiting for the memory system to go quiet after the search released its chunks */
} igdsp_io_report;
typedef struct igdsp_io_set igdsp_io_set;
int igdsp_io_alloc(igdsp_ctx *ctx, igdsp_io_buf *bufs, uint32_t n_bufs, size_t explore_limit_bytes,
                   igdsp_io_set **set, igdsp_io_report *report);
int igdsp_io_free(igdsp_ctx *ctx, igdsp_io_set *set);

int igdsp_copy_h2d(igdsp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int igdsp_copy_d2h(igdsp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int igdsp_dev_memset(igdsp_ctx *ctx, void *d_ptr, int value, size_t bytes);
int igdsp_sync(igdsp_ctx *ctx, void *stream);

/* ---- measurement helpers (HIP events on the launch stream; bench.py uses these
 * so timing does not depend on which stream torch considers current) ------------ */
int igdsp_timer_create(igdsp_ctx *ctx, void **timer);
int igdsp_timer_destroy(igdsp_ctx *ctx, void *timer);
int igdsp_timer_start(igdsp_ctx *ctx, void *timer, void *stream);
int igdsp_timer_stop(igdsp_ctx *ctx, void *timer, void *stream);
int igdsp_timer_elapsed_ms(igdsp_ctx *ctx, void *timer, float *ms); /* syncs on stop */

/* Read-only streaming-read calibration kernel (16 B/lane loads, xor-folded into
 * one word per workgroup): the "measured read-stream" the roofline is quoted
 * against besides the 8 TB/s nominal. */
int igdsp_stream_read(igdsp_ctx *ctx, const void *d_src, size_t bytes, uint64_t *d_sink, void *stream);

/* Placement probe.  On MI355X a stream that READS one large region of device memory and WRITES another runs ~13 %
 * faster than one that reads and writes the same region (regions are tens of GiB; inside one 80 GiB allocation the
 * bare read + record stream takes 0.218 ms across a region boundary and 0.252 ms within a region; the pure read rate
 * is the same everywhere; tools/placement_map.py grid).  This call times the bare read + record-store stream
 * (the meter kernel's traffic, no compute) reading d_in and writing d_out (bytes / 10 are written; NULL = a scratch
 * buffer allocated for the call), so a host can place its output buffers (records, PCM, re-encoded payload) a few
 * candidate distances away from its payload ring at start-up and keep the fastest.  Synchronous: 3 + reps launches
 * on `stream`, then waits. */
int igdsp_probe_placement(igdsp_ctx *ctx, const void *d_in, size_t bytes, void *d_out, uint32_t reps,
                          float *ms_per_launch, void *stream);

/* Kernel variant selection for experiments (0 = default tuned path).
 *   1 = one wavefront per channel-frame (40 lanes x dword), the literal north_star mapping
 *   2 = chunk64: one wavefront per 64 consecutive frames, 16 B/lane loads, 16 waves/CU (default for n == 160)
 *   3 = chunk64 with four super-chunks of lookahead per wave, 8 waves/CU (meter-only; experiment)
 *   4 = igdsp_roundtrip_peakhold through the compressor cell table (k_roundtrip_chunk64, the round-1 form) instead of
 *       the compressor folded into the expansion LUT; decode_meter as variant 0 */
int igdsp_set_variant(igdsp_ctx *ctx, int variant);

#ifdef __cplusplus
}
#endif
#endif /* IGDSP_H */
