// igdsp_host.cpp — host mirror of the reference's per-frame adapter/hook interface over the C ABI.
// See igdsp_host.h for the reference lines each piece stands for.  Own implementation throughout.
#include "igdsp_host.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

static RoIP_ED137 *theInstance_ = nullptr;

static long long now_ms()
{
    using namespace std::chrono;
    return duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
RoIP_ED137::RoIP_ED137() : inviteMode(SERVER), referenceTxQuirk(false), ed137Events(0), onValueChanged(nullptr), m_softPhoneID(1), ctx_(nullptr)
{
    std::memset(window, 0, sizeof window);
    for (int i = 0; i < 4; ++i) window[i].OutgoingRTPmin = 255;          // roip_ed137.h:745
    for (int i = 0; i < 4; ++i) {
        radio[i] = new trx();
        std::memset(radio[i], 0, sizeof(trx));
        radio[i]->call_id = -1;
    }
}

RoIP_ED137 *RoIP_ED137::create(int device, uint32_t max_calls)
{
    RoIP_ED137 *r = new (std::nothrow) RoIP_ED137();
    if (!r) return nullptr;
    // two metering channels per call: RX (IncomingRTP) and TX (OutgoingRTP)
    if (igdsp_create(&r->ctx_, device, max_calls * 2u) != IGDSP_OK) {
        delete r;
        return nullptr;
    }
    theInstance_ = r;
    return r;
}

RoIP_ED137 *RoIP_ED137::instance() { return theInstance_; }

RoIP_ED137::~RoIP_ED137()
{
    if (theInstance_ == this) theInstance_ = nullptr;
    igdsp_destroy(ctx_);
    for (int i = 0; i < 4; ++i) delete radio[i];
}

// call ids are mapped to metering channels through the C ABI's routing table; RX and TX of one
// call get distinct ids in that table: rx = call_id * 2, tx = call_id * 2 + 1.
static inline int32_t rx_key(int call_id) { return call_id * 2; }
static inline int32_t tx_key(int call_id) { return call_id * 2 + 1; }

int RoIP_ED137::bindRadio(int slot, int call_id)
{
    if (slot < 0 || slot > 3) return IGDSP_EINVAL;
    if (radio[slot]->call_id >= 0) {
        igdsp_unmap_call(ctx_, rx_key(radio[slot]->call_id));
        igdsp_unmap_call(ctx_, tx_key(radio[slot]->call_id));
    }
    radio[slot]->call_id = call_id;
    if (call_id < 0) return IGDSP_OK;
    int rc = igdsp_map_call(ctx_, rx_key(call_id), (uint32_t)(2 * slot));
    if (rc == IGDSP_OK) rc = igdsp_map_call(ctx_, tx_key(call_id), (uint32_t)(2 * slot + 1));
    return rc;
}

// setIncomingRTP reads callID, payload_buff and payload_bufSize from the adapter (roip_ed137.cpp:6549-6552)
// and only meters in SERVER mode (:6555).  The arithmetic itself now runs on the GPU at the next tick.
void RoIP_ED137::setIncomingRTP(tp_adapter *adapter)
{
    if (!adapter || inviteMode != SERVER) return;
    // the frame is staged under the adapter's current ED-137 word — what get_ed137_value(tp) would return at this moment
    // (ntohl(adapter->ed137_value), TransportAdapter.cpp:337-346): the flush gates the frame's fold into the call's window with it
    // when a gate mode is set (igdsp_set_gate_mode; PTT / SQU masks Functions.cpp:1136, 1160)
    (void)igdsp_set_ed137(ctx_, rx_key(adapter->callID), ntohl(adapter->ed137_value));
    (void)igdsp_on_rtp_frame(ctx_, rx_key(adapter->callID), adapter->last_rx_pt, adapter->payload_buff,
                             (uint32_t)adapter->payload_bufSize);
}

void RoIP_ED137::setOutgoingRTP(tp_adapter *adapter)
{
    if (!adapter || inviteMode != SERVER) return;
    const uint32_t n = (uint32_t)adapter->send_payload_bufSize;
    // reference: payloadbuf = tmp_payload_buf (the WHOLE packet) and the loop runs over its first n bytes
    // (roip_ed137.cpp:6505-6517).  Default here: meter the payload proper, 12 bytes in.
    const uint8_t *p = referenceTxQuirk ? adapter->tmp_payload_buf : adapter->tmp_payload_buf + IGDSP_RTP_HDR;
    (void)igdsp_on_rtp_frame(ctx_, tx_key(adapter->callID), adapter->last_tx_pt, p, n);
}

void RoIP_ED137::setIncomingED137Value(uint32_t, int) { ++ed137Events; }

int RoIP_ED137::tick(uint32_t *frames_done)
{
    int rc = igdsp_flush(ctx_, frames_done);
    if (rc != IGDSP_OK) return rc;
    for (int s = 0; s < 4; ++s) {
        trx *t = radio[s];
        if (t->call_id < 0) continue;
        igdsp_level lv;
        if (igdsp_poll(ctx_, (uint32_t)(2 * s), &lv) == IGDSP_OK && lv.frames) {
            t->IncomingRTP = lv.byte_mean; t->in_rms = lv.rms; t->in_peak = lv.peak; t->in_peak_hold = lv.peak_hold;
            t->in_percent = lv.percent; t->in_flags = lv.flags;
            if (onValueChanged) onValueChanged(t->call_id, 0, lv.percent);
        }
        if (igdsp_poll(ctx_, (uint32_t)(2 * s + 1), &lv) == IGDSP_OK && lv.frames) {
            t->OutgoingRTP = lv.byte_mean; t->out_rms = lv.rms; t->out_peak = lv.peak; t->out_peak_hold = lv.peak_hold;
            t->out_percent = lv.percent; t->out_flags = lv.flags;
            if (onValueChanged) onValueChanged(t->call_id, 1, lv.percent);
        }
    }
    return IGDSP_OK;
}

// ---------------------------------------------------------------------------------------------
// PTT-window logger.  keeplogAudioLevel: Functions.cpp:2126-2145.  createPTTEventDataLogger: reset on
// "pptTest_pressed" (:2155-2167), close + 10*log10 on "pptTest_released" (:2192-2222); message text :2169-2187, :2202-2215.
void RoIP_ED137::keeplogAudioLevel(int slot, double audioInLevel)
{
    if (slot < 0 || slot > 3) return;
    ptt_window &w = window[slot];
    if (!w.eventPttSQL_In_LoggingOn) return;
    const uint8_t out = radio[slot]->OutgoingRTP;
    w.level_in_count += 1;
    w.level_in = audioInLevel;
    w.level_in_av += audioInLevel;
    w.OutgoingRTPSum = (uint16_t)(w.OutgoingRTPSum + out);
    if (audioInLevel > w.level_in_max) w.level_in_max = audioInLevel;
    if (audioInLevel < w.level_in_min) w.level_in_min = audioInLevel;
    if (out > w.OutgoingRTPmax) w.OutgoingRTPmax = out;
    if (out < w.OutgoingRTPmin) w.OutgoingRTPmin = out;
}

static int ptt_json(char *json, size_t cap, int id, const char *ev, double a, double b, double c, const char *url, int x, int y, int z)
{
    // QString::arg(double) prints like %g (6 significant digits); uint8_t arguments promote to int.
    // The key "radioUrl " carries the reference's trailing blank.
    int n = std::snprintf(json, cap,
        "{\"menuID\"                       :\"PTTEventDataLogger\", \"softPhoneID\"                  :%d, "
        "\"Ptt\"                          :\"%s\", \"level_in_av\"                  :%g, "
        "\"level_in_max\"                 :%g, \"level_in_min\"                 :%g, "
        "\"radioUrl \"                    :\"%s\",\"OutgoingRTPAv\"                :%d, "
        "\"OutgoingRTPmax\"               :%d, \"OutgoingRTPmin\"               :%d }",
        id, ev, a, b, c, url ? url : "", x, y, z);
    return (n < 0 || (size_t)n >= cap) ? 0 : n;
}

int RoIP_ED137::createPTTEventDataLogger(int slot, const char *strEvent, const char *url, double audioInLevel, char *json, size_t cap)
{
    if (slot < 0 || slot > 3 || !strEvent || !json || cap == 0 || inviteMode != SERVER) return 0;
    ptt_window &w = window[slot];
    trx *r = radio[slot];
    json[0] = 0;
    if (std::strcmp(strEvent, "pptTest_pressed") == 0) {
        if (w.eventPttSQL_In_LoggingOn) return 0;
        w.eventPttSQL_In_LoggingOn = true;
        w.level_in_count = 0;
        w.level_in = 10 * std::log10(audioInLevel);
        w.level_in_av = 0; w.level_in_max = 0; w.level_in_min = 255;
        w.OutgoingRTPSum = 0; w.OutgoingRTPmax = 0; w.OutgoingRTPmin = 255;
        igdsp_reset_hold(ctx_, (uint32_t)(2 * slot));            // the per-frame device window opens with it
        igdsp_reset_hold(ctx_, (uint32_t)(2 * slot + 1));
        return ptt_json(json, cap, m_softPhoneID, strEvent, w.level_in, w.level_in, w.level_in, url, r->OutgoingRTP, r->OutgoingRTP, r->OutgoingRTP);
    }
    if (std::strcmp(strEvent, "pptTest_released") == 0) {
        if (!w.eventPttSQL_In_LoggingOn) return 0;
        w.eventPttSQL_In_LoggingOn = false;
        w.level_in_av = 10 * std::log10(w.level_in_av / w.level_in_count);
        w.level_in_max = 10 * std::log10(w.level_in_max);
        w.level_in_min = 10 * std::log10(w.level_in_min);
        w.OutgoingRTPav = w.level_in_count ? (uint8_t)(w.OutgoingRTPSum / w.level_in_count) : 0;   // the reference divides unguarded
        const int n = ptt_json(json, cap, m_softPhoneID, strEvent, w.level_in_av, w.level_in_max, w.level_in_min, url,
                               w.OutgoingRTPav, w.OutgoingRTPmax, w.OutgoingRTPmin);
        w.level_in_count = 0; w.level_in = 0; w.level_in_av = 0; w.level_in_max = 0; w.level_in_min = 255;
        return n;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
extern "C" {

int igdsp_host_keeplog(void *h, int slot, double audioInLevel)
{
    if (!h) return IGDSP_EINVAL;
    static_cast<RoIP_ED137 *>(h)->keeplogAudioLevel(slot, audioInLevel);
    return IGDSP_OK;
}

int igdsp_host_ptt_event(void *h, int slot, const char *strEvent, const char *url, double audioInLevel, char *json, size_t cap)
{
    return h ? static_cast<RoIP_ED137 *>(h)->createPTTEventDataLogger(slot, strEvent, url, audioInLevel, json, cap) : 0;
}

int igdsp_host_get_window(void *h, int slot, ptt_window *out)
{
    if (!h || !out || slot < 0 || slot > 3) return IGDSP_EINVAL;
    *out = static_cast<RoIP_ED137 *>(h)->window[slot];
    return IGDSP_OK;
}

int decodeRtp(void *pkt, custom_rtp_hdr **hdr)
{
    *hdr = reinterpret_cast<custom_rtp_hdr *>(pkt);     // a cast, like TransportAdapter.cpp:408-415
    return 0;
}

// RX: header parse, PT gate, payload copy, keep-alive bookkeeping, then the hooks.
void transport_rtp_cb(void *user_data, void *pkt, long size)
{
    tp_adapter *adapter = static_cast<tp_adapter *>(user_data);
    if (!adapter || !pkt || size < (long)IGDSP_RTP_HDR) return;
    custom_rtp_hdr *rtphdr = nullptr;
    decodeRtp(pkt, &rtphdr);
    const unsigned pt = rtphdr->pt;
    const long hdr = adapter->radiostatus ? (long)sizeof(custom_rtp_hdr) : (long)IGDSP_RTP_HDR;
    if (adapter->radiostatus && size >= (long)sizeof(custom_rtp_hdr) && (pt == 8 || pt == 0 || pt == 18 || pt == 123)) {
        adapter->ed137_value = rtphdr->ed137;                  // TransportAdapter.cpp:252-256
        adapter->payloadsize = rtphdr->length;
    }
    long payloadlen = size - hdr;
    // runt (shorter than its header): the reference's unsigned subtraction wraps, fails its `< 1024` guard and
    // returns (TransportAdapter.cpp:279-291); oversize: it guards with `< 1024` into a 256-byte buffer (:286) —
    // both are dropped here without touching the buffers
    if (payloadlen < 0 || payloadlen > (long)sizeof(adapter->payload_buff)) {
        adapter->r2sPacket = now_ms();
        return;
    }
    adapter->payload_bufSize = (size_t)payloadlen;
    std::memcpy(adapter->payload_buff, static_cast<const uint8_t *>(pkt) + hdr, (size_t)payloadlen);
    adapter->last_rx_pt = (uint8_t)pt;
    adapter->r2sPacket = now_ms();
    RoIP_ED137 *app = RoIP_ED137::instance();
    if (pt != 123) {
        if (adapter->stream_rtp_cb) adapter->stream_rtp_cb(adapter->stream_user_data, pkt, size);
        if (app) {
            app->setIncomingRTP(adapter);
            if (!adapter->rtpAudio) app->setIncomingED137Value(ntohl(adapter->ed137_value), adapter->callID);
        }
        adapter->rtpAudio = 1;
    } else {
        if (adapter->rtpAudio && app) app->setIncomingED137Value(ntohl(adapter->ed137_value), adapter->callID);
        adapter->rtpAudio = 0;
    }
}

// TX: the level / silence-probe part of transport_send_rtp (the ED-137 word assembly and the socket
// send stay with the softphone).  `pkt` is the pjmedia stream's 12-byte-header RTP packet.
int transport_send_rtp(tp_adapter *adapter, const void *pkt, size_t size)
{
    if (!adapter || !pkt || size < (size_t)IGDSP_RTP_HDR || size > sizeof(adapter->tmp_payload_buf)) return IGDSP_EINVAL;
    std::memcpy(adapter->tmp_payload_buf, pkt, size);          // whole packet, as TransportAdapter.cpp:654
    const uint8_t *b = adapter->tmp_payload_buf;
    if (size > 60) {                                           // TransportAdapter.cpp:657-673
        if (b[40] == b[50] && b[40] == b[60] && b[40] == 0xD5) { if (adapter->rtpFalse < 32767) adapter->rtpFalse++; }
        else adapter->rtpFalse = 0;
    }
    const unsigned pt = b[1] & 0x7F;
    adapter->last_tx_pt = (uint8_t)pt;
    if (pt != 123) {
        adapter->send_payload_bufSize = size - IGDSP_RTP_HDR;
        RoIP_ED137 *app = RoIP_ED137::instance();
        if (app) app->setOutgoingRTP(adapter);
    }
    return 0;
}

void *igdsp_host_create(int device, uint32_t max_calls) { return RoIP_ED137::create(device, max_calls); }
void igdsp_host_destroy(void *h) { delete static_cast<RoIP_ED137 *>(h); }

tp_adapter *igdsp_host_adapter_new(int call_id, int radiostatus)
{
    tp_adapter *a = new (std::nothrow) tp_adapter();
    if (!a) return nullptr;
    std::memset(a, 0, sizeof *a);
    a->callID = call_id;
    a->radiostatus = radiostatus;
    a->r2sPacket = now_ms();
    return a;
}

void igdsp_host_adapter_free(tp_adapter *a) { delete a; }
int igdsp_host_bind_radio(void *h, int slot, int call_id) { return h ? static_cast<RoIP_ED137 *>(h)->bindRadio(slot, call_id) : IGDSP_EINVAL; }

int igdsp_host_set_mode(void *h, int invite_mode, int reference_tx_quirk)
{
    if (!h) return IGDSP_EINVAL;
    static_cast<RoIP_ED137 *>(h)->inviteMode = invite_mode;
    static_cast<RoIP_ED137 *>(h)->referenceTxQuirk = reference_tx_quirk != 0;
    return IGDSP_OK;
}

int igdsp_host_tick(void *h, uint32_t *frames_done) { return h ? static_cast<RoIP_ED137 *>(h)->tick(frames_done) : IGDSP_EINVAL; }

int igdsp_host_get_trx(void *h, int slot, trx *out)
{
    if (!h || !out || slot < 0 || slot > 3) return IGDSP_EINVAL;
    *out = *static_cast<RoIP_ED137 *>(h)->radio[slot];
    return IGDSP_OK;
}

igdsp_ctx *igdsp_host_ctx(void *h) { return h ? static_cast<RoIP_ED137 *>(h)->ctx() : nullptr; }
uint32_t igdsp_host_ed137_events(void *h) { return h ? static_cast<RoIP_ED137 *>(h)->ed137Events : 0; }

// ---------------------------------------------------------------------------------------------
// FIFO producer for the reference's AudioMeter consumer (audiometer.cpp:11-34)
int igdsp_meter_fifo_open(const char *card, int timeout_ms)
{
    if (!card) return IGDSP_EINVAL;
    char path[256];
    if (std::snprintf(path, sizeof path, "/tmp/capturefifo%s", card) >= (int)sizeof path) return IGDSP_EINVAL;
    for (int waited = 0;; ++waited) {
        const int fd = ::open(path, O_WRONLY | O_NONBLOCK);          // fails with ENXIO until the reader has opened it
        if (fd >= 0) {
            ::fcntl(fd, F_SETFL, ::fcntl(fd, F_GETFL) & ~O_NONBLOCK);
            return fd;
        }
        if (waited >= timeout_ms) return IGDSP_ENOENT;
        ::usleep(1000);
    }
}

int igdsp_meter_fifo_write(int fd, int level)
{
    if (fd < 0) return IGDSP_EINVAL;
    char rec[32];                                                   // one record per read(in_fd, in, 32), NUL padded
    std::memset(rec, 0, sizeof rec);
    std::snprintf(rec, sizeof rec, "%d", level);
    return ::write(fd, rec, sizeof rec) == (ssize_t)sizeof rec ? IGDSP_OK : IGDSP_EDEVICE;
}

int igdsp_meter_fifo_close(int fd) { return (fd >= 0 && ::close(fd) == 0) ? IGDSP_OK : IGDSP_EINVAL; }

// ---------------------------------------------------------------------------------------------
// Recorder with the reference's writeRTPWav signature and byte-for-byte the same file:
// 44-byte header (tag 7, 2 "channels", 16 bit, sizes patched on stop) and every payload byte b
// written as the two bytes [b, 0x00] (WavWriter.cpp:63-156).  Buffered: one fwrite per frame
// instead of the reference's two per sample.
struct igdsp_wav { FILE *f; uint32_t data; };

static void le(uint8_t *p, uint32_t v, int n) { for (int i = 0; i < n; ++i) { p[i] = (uint8_t)v; v >>= 8; } }

void *igdsp_wav_start(const char *path, int rate)
{
    igdsp_wav *w = new (std::nothrow) igdsp_wav();
    if (!w) return nullptr;
    w->f = std::fopen(path, "wb");
    w->data = 0;
    if (!w->f) { delete w; return nullptr; }
    uint8_t h[44];
    std::memcpy(h, "RIFF", 4); le(h + 4, 0, 4); std::memcpy(h + 8, "WAVEfmt ", 8); le(h + 16, 16, 4);
    le(h + 20, 0x0007, 2); le(h + 22, 2, 2); le(h + 24, (uint32_t)rate, 4); le(h + 28, (uint32_t)rate * 4u, 4);
    le(h + 32, 4, 2); le(h + 34, 16, 2); std::memcpy(h + 36, "data", 4); le(h + 40, 0, 4);
    std::fwrite(h, 1, 44, w->f);
    return w;
}

int igdsp_wav_writeRTPWav(void *wv, const char *pktbuf, const char *payloadbuf, unsigned pktlen, unsigned payloadlen)
{
    (void)pktbuf; (void)pktlen;
    igdsp_wav *w = static_cast<igdsp_wav *>(wv);
    if (!w || !w->f || (!payloadbuf && payloadlen)) return IGDSP_EINVAL;
    uint8_t buf[2 * IGDSP_MAX_PAYLOAD];
    while (payloadlen) {
        unsigned n = payloadlen > IGDSP_MAX_PAYLOAD ? IGDSP_MAX_PAYLOAD : payloadlen;
        for (unsigned i = 0; i < n; ++i) { buf[2 * i] = (uint8_t)payloadbuf[i]; buf[2 * i + 1] = 0; }
        if (std::fwrite(buf, 1, 2 * n, w->f) != 2 * n) return IGDSP_EDEVICE;
        w->data += 2 * n; payloadbuf += n; payloadlen -= n;
    }
    return IGDSP_OK;
}

int igdsp_wav_stop(void *wv)
{
    igdsp_wav *w = static_cast<igdsp_wav *>(wv);
    if (!w) return IGDSP_EINVAL;
    uint8_t s[4];
    std::fseek(w->f, 4, SEEK_SET); le(s, 36 + w->data, 4); std::fwrite(s, 1, 4, w->f);
    std::fseek(w->f, 40, SEEK_SET); le(s, w->data, 4); std::fwrite(s, 1, 4, w->f);
    std::fclose(w->f);
    delete w;
    return IGDSP_OK;
}

}  // extern "C"
