/*
 * igdsp_oracle.c — TEST INFRASTRUCTURE ONLY (see igdsp_oracle.h for the parity
 * status of every function).  Plain scalar C, one obvious loop per function;
 * no SIMD, no tables where a formula is the definition.
 */
#define _GNU_SOURCE
#include "igdsp_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* ITU-T G.711 expansion.  mu-law: code is stored bit-inverted; 3-bit segment */
/* s, 4-bit step q; magnitude = ((2q+33) << s) - 33 in the 14-bit domain,     */
/* delivered left-justified in 16 bits (x4).  A-law: code XOR 0x55 (even-bit  */
/* inversion); segment 0 is linear (2q+1), segments >=1 are (2q+33) << (s-1), */
/* in the 13-bit domain, delivered x8.  Sign bit set = positive in both laws. */
/* pjmedia performs this step behind adapter->stream_rtp_cb                   */
/* (TransportAdapter.cpp:301); PT gate TransportAdapter.cpp:252.              */
/* ------------------------------------------------------------------------- */
int16_t orc_ulaw2lin(uint8_t code)
{
    unsigned u = (unsigned)(~code) & 0xFFu;
    int s = (int)((u >> 4) & 7u);
    int q = (int)(u & 15u);
    int mag14 = ((2 * q + 33) << s) - 33;
    int v = mag14 * 4;
    return (int16_t)((u & 0x80u) ? -v : v);
}

int16_t orc_alaw2lin(uint8_t code)
{
    unsigned a = (unsigned)code ^ 0x55u;
    int s = (int)((a >> 4) & 7u);
    int q = (int)(a & 15u);
    int mag13 = (s == 0) ? (2 * q + 1) : ((2 * q + 33) << (s - 1));
    int v = mag13 * 8;
    return (int16_t)((a & 0x80u) ? v : -v);
}

void orc_decode_table(int pt, int16_t out[256])
{
    for (int i = 0; i < 256; ++i)
        out[i] = (pt == 8) ? orc_alaw2lin((uint8_t)i) : orc_ulaw2lin((uint8_t)i);
}

/* index of the first segment whose upper bound holds `mag`; 8 = above all */
static int segment_of(int mag, int first_end)
{
    int end = first_end;
    for (int s = 0; s < 8; ++s) {
        if (mag <= end) return s;
        end = (end << 1) | 1;
    }
    return 8;
}

/* G.711 compression.  pjmedia performs this before transport_send_rtp
 * (TransportAdapter.cpp:635).  Two lineages, see include/igdsp.h. */
uint8_t orc_lin2ulaw(int16_t pcm, int variant)
{
    int v = pcm;
    unsigned flip;
    if (variant == 0) {                      /* SUN16: bias 0x84 in the 16-bit domain */
        int mag;
        if (v < 0) { mag = 0x84 - v; flip = 0x7Fu; } else { mag = v + 0x84; flip = 0xFFu; }
        int s = segment_of(mag, 0xFF);
        if (s >= 8) return (uint8_t)(0x7Fu ^ flip);
        return (uint8_t)((((unsigned)s << 4) | (((unsigned)mag >> (s + 3)) & 15u)) ^ flip);
    } else {                                 /* G191: 14-bit domain, clip 8159, bias 0x21 */
        int v14 = v >> 2;                    /* arithmetic shift: floors negatives */
        int mag;
        if (v14 < 0) { mag = -v14; flip = 0x7Fu; } else { mag = v14; flip = 0xFFu; }
        if (mag > 8159) mag = 8159;
        mag += 0x21;
        int s = segment_of(mag, 0x3F);
        if (s >= 8) return (uint8_t)(0x7Fu ^ flip);
        return (uint8_t)((((unsigned)s << 4) | (((unsigned)mag >> (s + 1)) & 15u)) ^ flip);
    }
}

uint8_t orc_lin2alaw(int16_t pcm, int variant)
{
    int v = pcm;
    unsigned flip;
    if (variant == 0) {                      /* SUN16: 16-bit domain, "-pcm-8", clamp at 0 */
        int mag;
        if (v >= 0) { mag = v; flip = 0xD5u; }
        else { mag = -v - 8; if (mag < 0) mag = 0; flip = 0x55u; }
        int s = segment_of(mag, 0xFF);
        if (s >= 8) return (uint8_t)(0x7Fu ^ flip);
        unsigned q = (s < 2) ? (((unsigned)mag >> 4) & 15u) : (((unsigned)mag >> (s + 3)) & 15u);
        return (uint8_t)((((unsigned)s << 4) | q) ^ flip);
    } else {                                 /* G191: 13-bit domain, "-pcm-1" */
        int v13 = v >> 3;
        int mag;
        if (v13 >= 0) { mag = v13; flip = 0xD5u; } else { mag = -v13 - 1; flip = 0x55u; }
        int s = segment_of(mag, 0x1F);
        if (s >= 8) return (uint8_t)(0x7Fu ^ flip);
        unsigned q = (s < 2) ? (((unsigned)mag >> 1) & 15u) : (((unsigned)mag >> s) & 15u);
        return (uint8_t)((((unsigned)s << 4) | q) ^ flip);
    }
}

/* ------------------------------------------------------------------------- */
/* roip_ed137.cpp:6557-6568 (RX) and 6511-6517 (TX): int accumulator over     */
/* payloadbuf[i] (a `const char*`; unsigned on the aarch64 product target),   */
/* integer division by payloadlen, truncation to uint8_t.  The `i = 4` skip   */
/* at :6561-6562 is overwritten by `for (i = 0; ...)` so no byte is skipped.  */
/* ------------------------------------------------------------------------- */
uint8_t orc_byte_mean(const uint8_t *payload, int payloadlen)
{
    int audioLevelSum = 0;
    if (payloadlen <= 0) return 0;           /* reference divides by zero here; we define 0 */
    for (int i = 0; i < payloadlen; i++) audioLevelSum += (unsigned char)payload[i];
    return (uint8_t)(audioLevelSum / payloadlen);
}

uint8_t orc_byte_mean_signed_char(const uint8_t *payload, int payloadlen)
{
    int audioLevelSum = 0;
    if (payloadlen <= 0) return 0;
    for (int i = 0; i < payloadlen; i++) audioLevelSum += (signed char)payload[i];
    return (uint8_t)(audioLevelSum / payloadlen);
}

/* audiometer.cpp:30-31: int(float((v*100.0)/30000.0)) */
int orc_percent(double level)
{
    return (int)(float)((level * 100.0) / 30000.0);
}

/* ------------------------------------------------------------------------- */
void orc_frame(const uint8_t *payload, int n, int pt, orc_frame_stats *st, int16_t *pcm)
{
    uint64_t sumsq = 0;
    int peak = 0;
    memset(st, 0, sizeof *st);
    if (n <= 0) { st->flags = 0x08; return; }
    for (int i = 0; i < n; ++i) {
        int x = (pt == 8) ? orc_alaw2lin(payload[i]) : orc_ulaw2lin(payload[i]);
        if (pcm) pcm[i] = (int16_t)x;
        sumsq += (uint64_t)((int64_t)x * x);
        int ax = x < 0 ? -x : x;
        if (ax > peak) peak = ax;
    }
    st->sumsq = sumsq;
    st->rms = (float)sqrt((double)sumsq / (double)n);   /* float64 reference, rounded once for storage */
    st->peak = (uint16_t)peak;
    st->byte_mean = orc_byte_mean(payload, n);
    uint8_t fl = 0;
    if (peak <= 8) fl |= 0x01;
    if (n > 48 && payload[28] == 0xD5 && payload[38] == 0xD5 && payload[48] == 0xD5) fl |= 0x02; /* TransportAdapter.cpp:657-660 */
    if (peak == ((pt == 8) ? 32256 : 32124)) fl |= 0x04;
    st->flags = fl;
}

void orc_decode_meter(const uint8_t *payload, const uint8_t *codec, const uint16_t *len,
                      uint32_t C, uint32_t F, uint32_t n,
                      orc_frame_stats *stats, int16_t *pcm, orc_aggregate *agg, uint32_t rank)
{
    uint64_t peak = 0;
    for (uint32_t f = 0; f < F; ++f)
        for (uint32_t c = 0; c < C; ++c) {
            size_t fi = (size_t)f * C + c;
            int l = len ? (int)len[fi] : (int)n;
            if (l > (int)n) l = (int)n;
            orc_frame_stats st;
            int16_t *p = pcm ? pcm + fi * n : NULL;
            orc_frame(payload + fi * n, l, codec[c], &st, p);
            if (p) for (uint32_t i = (uint32_t)l; i < n; ++i) p[i] = 0;
            stats[fi] = st;
            if (agg && l > 0) {
                agg->sumsq += st.sumsq; agg->samples += (uint64_t)l; agg->frames += 1;
                agg->n_silent += (st.flags & 1) ? 1 : 0; agg->n_clipped += (st.flags & 4) ? 1 : 0;
                agg->byte_mean_sum += st.byte_mean;
                if (st.peak > peak) peak = st.peak;
            }
        }
    if (agg && peak > agg->peak_slot[rank & 7]) agg->peak_slot[rank & 7] = peak;
}

void orc_encode(const int16_t *pcm, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                uint8_t *out, int variant)
{
    for (uint32_t f = 0; f < F; ++f)
        for (uint32_t c = 0; c < C; ++c) {
            size_t base = ((size_t)f * C + c) * n;
            for (uint32_t i = 0; i < n; ++i)
                out[base + i] = (codec[c] == 8) ? orc_lin2alaw(pcm[base + i], variant)
                                                : orc_lin2ulaw(pcm[base + i], variant);
        }
}

/* Functions.cpp:2155-2167 (reset on PTT press): count 0, sum 0, max 0, min 255 */
void orc_hold_reset(orc_chan_hold *hold, uint32_t C, const uint8_t *mask)
{
    for (uint32_t c = 0; c < C; ++c)
        if (!mask || mask[c]) { memset(&hold[c], 0, sizeof hold[c]); hold[c].level_min = 255; }
}

/* Functions.cpp:2126-2145 per frame while the window is open */
void orc_hold_update(const orc_frame_stats *stats, uint32_t C, uint32_t F, uint32_t n,
                     orc_chan_hold *hold, const uint8_t *gate)
{
    for (uint32_t f = 0; f < F; ++f)
        for (uint32_t c = 0; c < C; ++c) {
            const orc_frame_stats *s = &stats[(size_t)f * C + c];
            if (gate && !gate[c]) continue;
            if (s->flags & 0x08) continue;
            orc_chan_hold *h = &hold[c];
            h->count += 1;
            h->level_sum += s->byte_mean;
            h->samples += n;
            h->sumsq_acc += s->sumsq;
            if (s->peak > h->peak_hold) h->peak_hold = s->peak;
            if (s->byte_mean > h->level_max) h->level_max = s->byte_mean;
            if (s->byte_mean < h->level_min) h->level_min = s->byte_mean;
            h->n_silent += (s->flags & 1) ? 1 : 0;
            h->n_clipped += (s->flags & 4) ? 1 : 0;
        }
}

void orc_roundtrip_peakhold(const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                            uint8_t *out, orc_frame_stats *stats, orc_chan_hold *hold,
                            const uint8_t *gate, int variant)
{
    int16_t pcm[256];
    for (uint32_t f = 0; f < F; ++f)
        for (uint32_t c = 0; c < C; ++c) {
            size_t fi = (size_t)f * C + c;
            orc_frame(payload + fi * n, (int)n, codec[c], &stats[fi], pcm);
            for (uint32_t i = 0; i < n; ++i)
                out[fi * n + i] = (codec[c] == 8) ? orc_lin2alaw(pcm[i], variant) : orc_lin2ulaw(pcm[i], variant);
        }
    orc_hold_update(stats, C, F, n, hold, gate);
}

/* ------------------------------------------------------------------------- */
/* ED-137 gated window + consecutive-silence run (see igdsp_oracle.h)           */
/* ------------------------------------------------------------------------- */
static int orc_frame_gate(uint32_t mode, uint32_t ed137)
{
    const uint32_t ptt = (ed137 & 0xe0000000u) >> 29;      /* Functions.cpp:1136 */
    const uint32_t squ = (ed137 & 0x10000000u) >> 28;      /* Functions.cpp:1160 */
    switch (mode) {
    case 0: return 1;
    case 1: return squ != 0;
    case 2: return ptt != 0;
    default: return squ != 0 || ptt != 0;
    }
}

void orc_window_update(const orc_frame_stats *stats, const orc_rtp_info *info, const uint16_t *len,
                       uint32_t C, uint32_t F, uint32_t n, uint32_t gate_mode, uint32_t alarm,
                       orc_chan_hold *hold, const uint8_t *gate, orc_chan_probe *probe)
{
    if (alarm == 0) alarm = 500;                          /* TransportAdapter.cpp:666 */
    for (uint32_t f = 0; f < F; ++f)
        for (uint32_t c = 0; c < C; ++c) {
            const size_t fi = (size_t)f * C + c;
            const orc_frame_stats *s = &stats[fi];
            if (s->flags & 0x08) continue;                /* EMPTY: not metered, touches nothing */
            uint32_t l = len ? len[fi] : (info ? info[fi].payload_len : n);
            if (l > n) l = n;
            if (probe && l > 48) {                        /* TransportAdapter.cpp:657-673 */
                if (s->flags & 0x02) { if (++probe[c].run == alarm) probe[c].alarms += 1; }
                else probe[c].run = 0;
            }
            if (gate && !gate[c]) continue;
            if (!orc_frame_gate(gate_mode, info ? info[fi].ed137 : 0u)) continue;
            orc_chan_hold *h = &hold[c];                  /* keeplogAudioLevel, Functions.cpp:2126-2145 */
            h->count += 1;
            h->level_sum += s->byte_mean;
            h->samples += l;
            h->sumsq_acc += s->sumsq;
            if (s->peak > h->peak_hold) h->peak_hold = s->peak;
            if (s->byte_mean > h->level_max) h->level_max = s->byte_mean;
            if (s->byte_mean < h->level_min) h->level_min = s->byte_mean;
            h->n_silent += (s->flags & 1) ? 1 : 0;
            h->n_clipped += (s->flags & 4) ? 1 : 0;
        }
}

/* ------------------------------------------------------------------------- */
/* changeUplinkOrder (roip_ed137.cpp:6379-6499).  The reference fills packed   */
/* bit-field structs (GCC allocates fields from the least significant bit) and */
/* memcpy()s them out; here every output byte is written as the OR of its      */
/* fields at their bit positions.  `Input` is `const char*` = unsigned on the  */
/* product target.  Mode 4's S2_ is a 2-bit field assigned (b0 >> 5) & 0x04 —  */
/* the value 4 does not fit, so the field is always 0 (kept: bug-for-bug).     */
/* ------------------------------------------------------------------------- */
void orc_g726_reorder(const uint8_t *in, uint8_t *out, size_t n, int mode)
{
    if (mode == 1) {
        for (size_t k = 0; k < n; ++k) {
            unsigned b = in[k];
            out[k] = (uint8_t)((b << 6) | (b >> 6) | ((b & 0x30) >> 2) | ((b & 0x0C) << 2));
        }
    } else if (mode == 3) {
        for (size_t k = 0; k < n; ++k) { unsigned b = in[k]; out[k] = (uint8_t)((b >> 4) | (b << 4)); }
    } else if (mode == 2) {
        for (size_t o = 0; o + 3 <= n; o += 3) {
            unsigned V = (unsigned)in[o] | ((unsigned)in[o + 1] << 8) | ((unsigned)in[o + 2] << 16);
            unsigned S1 = V & 7, S2 = (V >> 3) & 7, S3 = (V >> 7) & 3, S3_ = (V >> 6) & 1, S4 = (V >> 9) & 7;
            unsigned S5 = (V >> 12) & 7, S6 = (V >> 17) & 1, S6_ = (V >> 15) & 3, S7 = (V >> 18) & 7, S8 = (V >> 21) & 7;
            out[o] = (uint8_t)(S3 | (S2 << 2) | (S1 << 5));             /* S3:2 S2:3 S1:3 */
            out[o + 1] = (uint8_t)(S6 | (S5 << 1) | (S4 << 4) | (S3_ << 7)); /* S6:1 S5:3 S4:3 S3_:1 */
            out[o + 2] = (uint8_t)(S8 | (S7 << 3) | (S6_ << 6));        /* S8:3 S7:3 S6_:2 */
        }
    } else if (mode == 4) {
        for (size_t o = 0; o + 5 <= n; o += 5) {
            unsigned t0 = in[o], t1 = in[o + 1], t2 = in[o + 2], t3 = in[o + 3], t4 = in[o + 4];
            unsigned S1 = t0 & 0x1F, S2 = ((t1 << 1) | (t0 >> 7)) & 0x07;
            unsigned S2_ = ((t0 >> 5) & 0x04) & 0x03;                     /* 2-bit field: always 0 */
            unsigned S3 = (t1 >> 2) & 0x1F, S4 = (t2 >> 3) & 0x01, S4_ = ((t2 << 1) | (t1 >> 7)) & 0x0F;
            unsigned S5 = ((t3 << 3) | (t2 >> 5)) & 0x0F, S5_ = (t2 >> 4) & 0x01, S6 = (t3 >> 1) & 0x1F;
            unsigned S7 = (t4 >> 1) & 0x03, S7_ = ((t4 << 2) | (t3 >> 6)) & 0x07, S8 = (t4 >> 3) & 0x1F;
            out[o] = (uint8_t)(S2 | (S1 << 3));                          /* S2:3 S1:5 */
            out[o + 1] = (uint8_t)(S4 | (S3 << 1) | (S2_ << 6));        /* S4:1 S3:5 S2_:2 */
            out[o + 2] = (uint8_t)(S5 | (S4_ << 4));                    /* S5:4 S4_:4 */
            out[o + 3] = (uint8_t)(S7 | (S6 << 2) | (S5_ << 7));        /* S7:2 S6:5 S5_:1 */
            out[o + 4] = (uint8_t)(S8 | (S7_ << 5));                    /* S8:5 S7_:3 */
        }
    }
}

/* ------------------------------------------------------------------------- */
/* transport_rtp_cb (TransportAdapter.cpp:240-292): the header is the first    */
/* 12 bytes (plain SIP) or 20 bytes (radio: custom_rtp_hdr, ed137_rtp.h:22-48) */
/* of the packet; payloadlen = size - header; PT is the low 7 bits of byte 1;  */
/* the ED-137 word is bytes 16..19 in network order (get_ed137_value does      */
/* ntohl, :337-346) and is only taken when pt is 8, 0, 18 or 123 (:252).       */
/* pt == 123 is a keep-alive: never handed to setIncomingRTP (:299,308).       */
/* ------------------------------------------------------------------------- */
void orc_depayload(const uint8_t *packets, const uint16_t *sizes, const uint8_t *radio, uint32_t C, uint32_t F,
                   uint32_t stride, uint32_t n, uint8_t *payload, uint16_t *len, orc_rtp_info *info)
{
    for (uint32_t f = 0; f < F; ++f)
        for (uint32_t c = 0; c < C; ++c) {
            size_t fi = (size_t)f * C + c;
            const uint8_t *pkt = packets + fi * stride;
            uint32_t size = sizes ? sizes[fi] : stride;
            if (size > stride) size = stride;
            uint32_t hdr = radio[c] ? 20u : 12u;
            orc_rtp_info in; memset(&in, 0, sizeof in);
            uint8_t *out = payload + fi * n;
            memset(out, 0, n);
            len[fi] = 0;
            if (size < hdr) { in.flags = 0x40; if (size >= 2) in.pt = pkt[1] & 0x7F; info[fi] = in; continue; }
            unsigned pt = pkt[1] & 0x7Fu;
            in.pt = (uint8_t)pt;
            in.payload_len = (uint16_t)(size - hdr);
            if ((pkt[0] >> 6) == 2) in.flags |= 0x01;
            if (pkt[0] & 0x10) in.flags |= 0x02;
            if (pkt[1] & 0x80) in.flags |= 0x04;
            if (radio[c]) {
                if (pt == 8 || pt == 0 || pt == 18 || pt == 123)
                    in.ed137 = ((uint32_t)pkt[16] << 24) | ((uint32_t)pkt[17] << 16) | ((uint32_t)pkt[18] << 8) | pkt[19];
                if ((pkt[0] & 0x10) && pkt[12] == 0x01 && pkt[13] == 0x67 && pkt[14] == 0x00 && pkt[15] == 0x01) in.flags |= 0x08;
            }
            if (pt == 123) in.flags |= 0x10;
            uint32_t pl = size - hdr;
            if (pl > n) in.flags |= 0x80;
            else if ((pt == 0 || pt == 8) && pl > 0) {
                in.flags |= 0x20;
                memcpy(out, pkt + hdr, pl);
                len[fi] = (uint16_t)pl;
            }
            info[fi] = in;
        }
}

/* ------------------------------------------------------------------------- */
uint64_t orc_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_gen_uniform(uint8_t *out, uint64_t n_bytes, uint64_t seed, uint64_t first_byte)
{
    for (uint64_t k = 0; k < n_bytes; ++k) {
        uint64_t g = first_byte + k;
        out[k] = (uint8_t)(orc_splitmix64(seed + (g >> 3)) >> (8 * (g & 7)));
    }
}

void orc_gen_speech(uint8_t *out, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                    uint64_t seed, uint32_t first_channel, int variant)
{
    const double w1 = 2.0 * 3.14159265358979323846 * 440.0 / 8000.0;
    const double w2 = 2.0 * 3.14159265358979323846 * 1000.0 / 8000.0;
    for (uint32_t f = 0; f < F; ++f)
        for (uint32_t c = 0; c < C; ++c) {
            uint32_t gc = first_channel + c;
            double amp = 1000.0 * (1 + (gc % 30));
            size_t base = ((size_t)f * C + c) * n;
            for (uint32_t i = 0; i < n; ++i) {
                uint64_t t = (uint64_t)f * n + i;
                uint64_t g = ((uint64_t)gc << 32) + t;
                int noise = (int)(orc_splitmix64(seed + g) & 0xFF) - 128;
                long v = lrint(amp * (0.6 * sin(w1 * (double)t) + 0.4 * sin(w2 * (double)t))) + noise;
                if (v > 32767) v = 32767;
                if (v < -32768) v = -32768;
                out[base + i] = (codec[c] == 8) ? orc_lin2alaw((int16_t)v, variant) : orc_lin2ulaw((int16_t)v, variant);
            }
        }
}

/* ------------------------------------------------------------------------- */
/* WavWriter.cpp:63-113 start(): "RIFF", total size, "WAVE", "fmt ", 16,       */
/* tag (fwrite length is the macro WAVE_FORMAT_ADPCM == 2, :95), channels 2,   */
/* rate, rate*16/8*2, block align 4, bits 16, "data" (first 4 bytes of the     */
/* literal "data "), data length; stop() :115-133 patches total-8 and data     */
/* length.  wav_write :136-156 emits each payload byte b as `channels` (=2)    */
/* little-endian bytes of the word b, i.e. [b, 0x00].                          */
/* ------------------------------------------------------------------------- */
static void put_le(uint8_t *p, uint32_t v, int nbytes) { for (int i = 0; i < nbytes; ++i) { p[i] = (uint8_t)v; v >>= 8; } }

size_t orc_wav_header(uint8_t out[44], uint32_t rate, uint32_t data_bytes)
{
    memcpy(out + 0, "RIFF", 4);
    put_le(out + 4, 36 + data_bytes, 4);          /* total - (totaloffset + 4), totaloffset = 4 */
    memcpy(out + 8, "WAVE", 4);
    memcpy(out + 12, "fmt ", 4);
    put_le(out + 16, 16, 4);
    put_le(out + 20, 0x0007, 2);                  /* WAVE_FORMAT_MULAW, Codecs.h:34; WavWriter.cpp:71 */
    put_le(out + 22, 2, 2);                       /* channels = 2, WavWriter.cpp:72 */
    put_le(out + 24, rate, 4);
    put_le(out + 28, rate * 16 / 8 * 2, 4);
    put_le(out + 32, 16 / 8 * 2, 2);
    put_le(out + 34, 16, 2);
    memcpy(out + 36, "data", 4);
    put_le(out + 40, data_bytes, 4);              /* total - (dataoffset + 4), dataoffset = 40 */
    return 44;
}

void orc_wav_expand(const uint8_t *payload, uint32_t n, uint8_t *out)
{
    for (uint32_t i = 0; i < n; ++i) { out[2 * i] = payload[i]; out[2 * i + 1] = 0; }
}

/* ------------------------------------------------------------------------- */
/* PTT-window logger, field-for-field after trx (roip_ed137.h:719-745)         */
/* ------------------------------------------------------------------------- */
void orc_ptt_init(orc_ptt_logger *l)
{
    memset(l, 0, sizeof *l);
    l->OutgoingRTPmin = 255;                      /* roip_ed137.h:745 */
}

void orc_ptt_pressed(orc_ptt_logger *l, double audioInLevel)   /* Functions.cpp:2155-2167 */
{
    if (!l->logging_on) {
        l->logging_on = 1;
        l->level_in_count = 0;
        l->level_in = 10 * log10(audioInLevel);
        l->level_in_av = 0; l->level_in_max = 0; l->level_in_min = 255;
        l->OutgoingRTPSum = 0; l->OutgoingRTPmax = 0; l->OutgoingRTPmin = 255;
    }
}

void orc_ptt_keeplog(orc_ptt_logger *l, double audioInLevel)   /* Functions.cpp:2126-2145 */
{
    if (!l->logging_on) return;
    l->level_in_count += 1;
    l->level_in = audioInLevel;
    l->level_in_av += audioInLevel;
    l->OutgoingRTPSum = (uint16_t)(l->OutgoingRTPSum + l->OutgoingRTP);
    if (audioInLevel > l->level_in_max) l->level_in_max = audioInLevel;
    if (audioInLevel < l->level_in_min) l->level_in_min = audioInLevel;
    if (l->OutgoingRTP > l->OutgoingRTPmax) l->OutgoingRTPmax = l->OutgoingRTP;
    if (l->OutgoingRTP < l->OutgoingRTPmin) l->OutgoingRTPmin = l->OutgoingRTP;
}

void orc_ptt_released(orc_ptt_logger *l)                        /* Functions.cpp:2192-2200 */
{
    if (!l->logging_on) return;
    l->logging_on = 0;
    l->level_in_av = 10 * log10(l->level_in_av / l->level_in_count);
    l->level_in_max = 10 * log10(l->level_in_max);
    l->level_in_min = 10 * log10(l->level_in_min);
    l->OutgoingRTPav = l->level_in_count ? (uint8_t)(l->OutgoingRTPSum / l->level_in_count) : 0;
}

/* ------------------------------------------------------------------------- */
/* cpu_baseline timing: table decode (as pjmedia does) is NOT used here — the  */
/* baseline is the scalar oracle itself, one frame at a time, -O2.             */
/* ------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *payload; const uint8_t *codec; uint32_t C, F, n, c0, c1; int reps;
    orc_frame_stats *stats; uint8_t *bm; int mode; const int32_t *calls;
} job_t;

static int16_t g_tab[2][256];
static int g_tab_ready;

/* B1 "through the single-frame shim" (BASELINE.md section 2): the CPU counterpart of igdsp_on_rtp_frame — the inputs
 * setIncomingRTP reads (callID, payload pointer, payload length, roip_ed137.cpp:6549-6552) plus the RTP PT
 * (TransportAdapter.cpp:252), ONE frame per call: route the call id to its slot (the reference's if-chain,
 * roip_ed137.cpp:6570-6585, as a table), table decode + sum x^2 + peak + byte mean + sqrt, leave the record in the slot.
 * Reached through a volatile function pointer so that the per-frame call is a real call, as it is across a .so boundary. */
typedef struct { const int32_t *slot_of_call; uint32_t n_calls; orc_frame_stats *slots; } orc_shim;

static int orc_shim_on_rtp_frame(orc_shim *s, int32_t call_id, uint8_t pt, const uint8_t *payload, uint32_t n)
{
    if (pt != 0 && pt != 8) return 0;
    if (call_id < 0 || (uint32_t)call_id >= s->n_calls) return -2;
    const int32_t slot = s->slot_of_call[call_id];
    if (slot < 0) return -2;
    if (n == 0) return 0;
    const int16_t *tab = g_tab[pt == 8];
    uint64_t ss = 0; int peak = 0; int bs = 0;
    for (uint32_t i = 0; i < n; ++i) {
        int x = tab[payload[i]];
        ss += (uint64_t)((int64_t)x * x);
        int ax = x < 0 ? -x : x; if (ax > peak) peak = ax;
        bs += payload[i];
    }
    orc_frame_stats *st = &s->slots[slot];
    st->sumsq = ss; st->rms = (float)sqrt((double)ss / n); st->peak = (uint16_t)peak;
    st->byte_mean = (uint8_t)(bs / (int)n);
    st->flags = (uint8_t)((peak <= 8) | ((peak == (pt == 8 ? 32256 : 32124)) << 2));
    return 0;
}
static int (*volatile g_frame_entry)(orc_shim *, int32_t, uint8_t, const uint8_t *, uint32_t) = orc_shim_on_rtp_frame;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    if (j->mode == 2) {          /* one call per frame; the record slots of this thread's channels are j->stats[c] */
        orc_shim shim = { j->calls, j->C, j->stats };
        for (int r = 0; r < j->reps; ++r)
            for (uint32_t f = 0; f < j->F; ++f)
                for (uint32_t c = j->c0; c < j->c1; ++c)
                    (void)g_frame_entry(&shim, (int32_t)c, j->codec[c], j->payload + ((size_t)f * j->C + c) * j->n, j->n);
        return NULL;
    }
    for (int r = 0; r < j->reps; ++r)
        for (uint32_t f = 0; f < j->F; ++f)
            for (uint32_t c = j->c0; c < j->c1; ++c) {
                size_t fi = (size_t)f * j->C + c;
                const uint8_t *p = j->payload + fi * j->n;
                if (j->mode == 1) { j->bm[fi] = orc_byte_mean(p, (int)j->n); continue; }
                /* B1 of BASELINE.md: scalar 256-entry-table decode + int64 sum x^2 + peak + sqrt */
                const int16_t *tab = g_tab[j->codec[c] == 8];
                uint64_t ss = 0; int peak = 0; int bs = 0;
                for (uint32_t i = 0; i < j->n; ++i) {
                    int x = tab[p[i]];
                    ss += (uint64_t)((int64_t)x * x);
                    int ax = x < 0 ? -x : x; if (ax > peak) peak = ax;
                    bs += p[i];
                }
                orc_frame_stats *st = &j->stats[fi];
                st->sumsq = ss; st->rms = (float)sqrt((double)ss / j->n); st->peak = (uint16_t)peak;
                st->byte_mean = (uint8_t)(bs / (int)j->n);
                st->flags = (uint8_t)((peak <= 8) | ((peak == (j->codec[c] == 8 ? 32256 : 32124)) << 2));
            }
    return NULL;
}

static double run_threads(job_t proto, int threads)
{
    if (!g_tab_ready) { orc_decode_table(0, g_tab[0]); orc_decode_table(8, g_tab[1]); g_tab_ready = 1; }
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > proto.C) threads = (int)proto.C;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; ++t) {
        jobs[t] = proto;
        jobs[t].c0 = (uint32_t)((uint64_t)proto.C * t / threads);
        jobs[t].c1 = (uint32_t)((uint64_t)proto.C * (t + 1) / threads);
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

double orc_time_decode_meter(const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                             int threads, int reps, orc_frame_stats *stats)
{
    job_t j = { payload, codec, C, F, n, 0, 0, reps, stats, NULL, 0, NULL };
    return run_threads(j, threads);
}

/* the same workload, ONE FRAME PER CALL through the single-frame shim (call id c -> slot c); slots = C records */
double orc_time_single_frame(const uint8_t *payload, const uint8_t *codec, uint32_t C, uint32_t F, uint32_t n,
                             int threads, int reps, orc_frame_stats *slots)
{
    int32_t *calls = (int32_t *)malloc(sizeof(int32_t) * (size_t)C);
    for (uint32_t c = 0; c < C; ++c) calls[c] = (int32_t)c;
    job_t j = { payload, codec, C, F, n, 0, 0, reps, slots, NULL, 2, calls };
    const double t = run_threads(j, threads);
    free(calls);
    return t;
}

double orc_time_byte_mean(const uint8_t *payload, uint32_t C, uint32_t F, uint32_t n, int threads, int reps,
                          uint8_t *out)
{
    job_t j = { payload, NULL, C, F, n, 0, 0, reps, NULL, out, 1, NULL };
    return run_threads(j, threads);
}
