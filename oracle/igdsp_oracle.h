/*
 * igdsp_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the hot path, used as the checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * igate4xsoftphonedsp_amd/ may include, link or call this.
 *
 * Parity status:
 *  - byte-mean "audioLevel", WAV expansion, PTT-window logger, percent scale:
 *    restated from reference source, file:line cited at each function.
 *  - G.711 decode/encode: NOT in the reference tree — performed by third-party
 *    PJSIP/pjmedia (pjmedia/src/pjmedia/g711.c + alaw_ulaw.c; version unpinned,
 *    linked by bare name iGate4xSoftphoneDSP.pro:66-82; hint "pjsip 2.6"
 *    TransportAdapter.cpp:245).  The reference holds no tests or vectors for it
 *    => "parity unpinned" against pjmedia.  Decode is pinned against ITU-T G.711
 *    via SHA-256 KATs + CPython audioop fixtures (tests/golden/); encode variant
 *    G191 is pinned exhaustively against audioop; variant SUN16 is pinned on the
 *    closed set enc(dec(c)) and on monotonicity / error-bound properties only.
 *    Which of the two pjmedia carries is UNVERIFIED (pjmedia is absent here).
 *  - RMS / peak: not computed anywhere in the reference (audiometer.cpp reads
 *    ASCII levels from a FIFO); definition adopted in SURVEY.md 8(a5).
 */
#ifndef IGDSP_ORACLE_H
#define IGDSP_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* same layouts as include/igdsp.h (tests assert sizeof/offset equality) */
typedef struct { uint64_t sumsq; float rms; uint16_t peak; uint8_t byte_mean; uint8_t flags; } orc_frame_stats;
typedef struct {
    uint64_t sumsq_acc; uint32_t count; uint32_t level_sum; uint32_t samples;
    uint16_t peak_hold; uint8_t level_max; uint8_t level_min; uint32_t n_silent; uint32_t n_clipped;
} orc_chan_hold;
typedef struct {   /* one 128-byte line per counter, as igdsp_aggregate */
    uint64_t sumsq, pad0[15], samples, pad1[15], frames, pad2[15], n_silent, pad3[15], n_clipped, pad4[15],
             byte_mean_sum, pad5[15], peak_slot[8], pad6[8];
} orc_aggregate;

/* G.711 (ITU-T G.711; restated from the standard's segment definition) */
int16_t orc_ulaw2lin(uint8_t code);
int16_t orc_alaw2lin(uint8_t code);
uint8_t orc_lin2ulaw(int16_t pcm, int variant);   /* variant 0 = SUN16, 1 = G191 */
uint8_t orc_lin2alaw(int16_t pcm, int variant);
void    orc_decode_table(int pt, int16_t out[256]);

/* roip_ed137.cpp:6564-6568 / 6513-6517 with unsigned-char semantics (aarch64 target) */
uint8_t orc_byte_mean(const uint8_t *payload, int payloadlen);
/* the same loop compiled for a signed-char host (x86), for the documentation test */
uint8_t orc_byte_mean_signed_char(const uint8_t *payload, int payloadlen);
/* audiometer.cpp:30-31 */
int orc_percent(double level);

/* one frame: decode + stats (+ optional PCM) */
void orc_frame(const uint8_t *payload, int n, int pt, orc_frame_stats *st, int16_t *pcm /*nullable*/);

/* batched [F][C][n]; len nullable; pcm nullable; agg nullable (ADDs, peak into slot[rank]) */
void orc_decode_meter(const uint8_t *payload, const uint8_t *codec, const uint16_t *len,
                      uint32_t C, uint32_t F, uint32_t n,
                      orc_frame_stats *stats, int16_t *pcm, orc_aggregate *agg, uint32_t rank);
void orc_encode(const int16_t *pcm, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                uint8_t *out, int variant);
void orc_hold_reset(orc_chan_hold *hold, uint32_t C, const uint8_t *mask);
void orc_hold_update(const orc_frame_stats *stats, uint32_t C, uint32_t F, uint32_t n,
                     orc_chan_hold *hold, const uint8_t *gate);
void orc_roundtrip_peakhold(const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                            uint8_t *out, orc_frame_stats *stats, orc_chan_hold *hold,
                            const uint8_t *gate, int variant);

/* TransportAdapter.cpp:240-292 restated: header parse + payload copy, batched */
typedef struct { uint32_t ed137; uint16_t payload_len; uint8_t pt; uint8_t flags; } orc_rtp_info;
void orc_depayload(const uint8_t *packets, const uint16_t *sizes, const uint8_t *radio, uint32_t C, uint32_t F,
                   uint32_t stride, uint32_t n, uint8_t *payload, uint16_t *len, orc_rtp_info *info);

/* The ED-137 gated window (last clause), restated from the reference's own pieces:
 *   window fold      keeplogAudioLevel, Functions.cpp:2126-2145 (count / sum / max / min while the window is open)
 *   gate bits        get_IPRadioPttStatus  (ed137 & 0xe0000000) >> 29, Functions.cpp:1136;
 *                    get_IPRadioSquelch    (ed137 & 0x10000000) >> 28, Functions.cpp:1160
 *   silence run      adapter->rtpFalse, TransportAdapter.cpp:657-673: += 1 when bytes 40 / 50 / 60 of the TX packet are 0xD5
 *                    (payload bytes 28 / 38 / 48 = IGDSP_FLAG_PROBE_D5), = 0 when they are not, untouched when the packet is
 *                    too short to hold them (size > 60 <=> payload length > 48); "alarm" when it reaches exactly 500
 * gate_mode 0 always, 1 SQU, 2 PTT type != 0, 3 either.  info nullable (ED-137 word 0; length from len or n); len nullable. */
typedef struct { uint32_t run, alarms; } orc_chan_probe;
void orc_window_update(const orc_frame_stats *stats, const orc_rtp_info *info, const uint16_t *len,
                       uint32_t C, uint32_t F, uint32_t n, uint32_t gate_mode, uint32_t alarm,
                       orc_chan_hold *hold, const uint8_t *gate, orc_chan_probe *probe /*nullable*/);

/* roip_ed137.cpp:6379-6499 changeUplinkOrder restated (unsigned-char target, GCC LSB-first bit-fields) */
void orc_g726_reorder(const uint8_t *in, uint8_t *out, size_t n, int mode);

/* synthetic data */
uint64_t orc_splitmix64(uint64_t x);
void orc_gen_uniform(uint8_t *out, uint64_t n_bytes, uint64_t seed, uint64_t first_byte);
/* D-speech: two-tone + noise per channel, encoded with the oracle encoder */
void orc_gen_speech(uint8_t *out, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                    uint64_t seed, uint32_t first_channel, int variant);

/* WavWriter.cpp:63-156 restated: header bytes and the [b,0x00] expansion */
size_t orc_wav_header(uint8_t out[44], uint32_t rate, uint32_t data_bytes);
void   orc_wav_expand(const uint8_t *payload, uint32_t n, uint8_t *out /* 2n */);

/* Functions.cpp:2126-2145 + 2155-2167 + 2192-2200: PTT-window level logger */
typedef struct {
    int    logging_on;       /* eventPttSQL_In_LoggingOn */
    int    level_in_count;
    double level_in, level_in_av, level_in_max, level_in_min;
    uint8_t  OutgoingRTP;    /* current byte-mean level */
    uint16_t OutgoingRTPSum; /* wraps mod 65536 like the reference's uint16_t */
    uint8_t  OutgoingRTPav, OutgoingRTPmax, OutgoingRTPmin;
} orc_ptt_logger;
void orc_ptt_init(orc_ptt_logger *l);
void orc_ptt_pressed(orc_ptt_logger *l, double audioInLevel);
void orc_ptt_keeplog(orc_ptt_logger *l, double audioInLevel);
void orc_ptt_released(orc_ptt_logger *l);

/* cpu_baseline: time `reps` passes of decode+meter over [F][C][n] on `threads`
 * pthreads (static channel partition); returns seconds for all reps. */
double orc_time_decode_meter(const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                             int threads, int reps, orc_frame_stats *stats);
/* B1 of BASELINE.md section 2 as written: one frame per call through a single-frame entry shaped like igdsp_on_rtp_frame
 * (call id -> slot routing, table decode, sum x^2, peak, byte mean, sqrt); slots = C records (each call's newest). */
double orc_time_single_frame(const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                             int threads, int reps, orc_frame_stats *slots);
double orc_time_byte_mean(const uint8_t *payload, uint32_t C, uint32_t F, uint32_t n, int threads, int reps,
                          uint8_t *out);
#ifdef __cplusplus
}
#endif
#endif
