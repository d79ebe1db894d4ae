// CPU sanitizer harness (ASan + UBSan): drives the oracle (the checker) and the C++ host mirror's GPU-free entry
// points with randomised and hostile inputs.  Built and run by tests/test_sanitize_cpu.py; test infrastructure only.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "igdsp_host.h"
#include "igdsp_oracle.h"

static uint64_t rng_state = 0x20241218ull;
static uint32_t rnd() { rng_state = orc_splitmix64(rng_state); return (uint32_t)(rng_state >> 32); }

static void oracle_pass()
{
    int16_t tab[256];
    orc_decode_table(0, tab); orc_decode_table(8, tab);
    unsigned acc = 0;
    for (int v = -32768; v <= 32767; ++v)
        for (int variant = 0; variant < 2; ++variant) acc += orc_lin2ulaw((int16_t)v, variant) + orc_lin2alaw((int16_t)v, variant);
    const uint32_t Cs[] = {1, 3, 64}, Fs[] = {1, 5}, ns[] = {1, 3, 24, 159, 160, 256};
    for (uint32_t C : Cs) for (uint32_t F : Fs) for (uint32_t n : ns) {
        std::vector<uint8_t> pl((size_t)C * F * n), codec(C), out((size_t)C * F * n), gate(C);
        std::vector<uint16_t> len((size_t)C * F);
        orc_gen_uniform(pl.data(), pl.size(), rnd(), 0);
        for (auto &c : codec) c = (rnd() & 1) ? 8 : 0;
        for (auto &g : gate) g = rnd() & 1;
        for (auto &l : len) l = (uint16_t)(rnd() % (n + 40));          // includes lengths past n (clamped by the oracle)
        std::vector<orc_frame_stats> st((size_t)C * F);
        std::vector<int16_t> pcm((size_t)C * F * n);
        orc_aggregate agg; memset(&agg, 0, sizeof agg);
        orc_decode_meter(pl.data(), codec.data(), len.data(), C, F, n, st.data(), pcm.data(), &agg, 7);
        orc_decode_meter(pl.data(), codec.data(), nullptr, C, F, n, st.data(), nullptr, nullptr, 0);
        for (int variant = 0; variant < 2; ++variant) orc_encode(pcm.data(), codec.data(), C, F, n, out.data(), variant);
        std::vector<orc_chan_hold> hold(C);
        orc_hold_reset(hold.data(), C, nullptr);
        orc_hold_update(st.data(), C, F, n, hold.data(), gate.data());
        orc_roundtrip_peakhold(pl.data(), codec.data(), C, F, n, out.data(), st.data(), hold.data(), nullptr, 0);
        orc_roundtrip_peakhold(pl.data(), codec.data(), C, F, n, out.data(), st.data(), hold.data(), gate.data(), 1);
        // the ED-137 gated window over the same records: every gate mode, info / len given or not, a tiny alarm length
        std::vector<orc_rtp_info> winfo((size_t)C * F);
        for (auto &i : winfo) { i.ed137 = rnd(); i.payload_len = (uint16_t)(rnd() % (n + 20)); i.pt = 0; i.flags = 0; }
        std::vector<orc_chan_probe> probe(C);
        memset(probe.data(), 0, C * sizeof(orc_chan_probe));
        for (uint32_t mode = 0; mode < 4; ++mode) {
            orc_window_update(st.data(), winfo.data(), nullptr, C, F, n, mode, 2, hold.data(), gate.data(), probe.data());
            orc_window_update(st.data(), nullptr, len.data(), C, F, n, mode, 0, hold.data(), nullptr, nullptr);
        }
        std::vector<orc_frame_stats> slots(C);
        acc += (unsigned)(orc_time_single_frame(pl.data(), codec.data(), C, F, n, 3, 1, slots.data()) >= 0.0) + probe[0].run;
        acc += st[0].peak + agg.frames;
    }
    for (uint32_t stride : {24u, 64u, 180u, 276u}) {
        const uint32_t C = 7, F = 9, n = stride > 200 ? 256u : 160u;
        std::vector<uint8_t> pk((size_t)C * F * stride), radio(C), pay((size_t)C * F * n);
        std::vector<uint16_t> sizes((size_t)C * F), len((size_t)C * F);
        std::vector<orc_rtp_info> info((size_t)C * F);
        orc_gen_uniform(pk.data(), pk.size(), rnd(), 0);
        for (auto &r : radio) r = rnd() & 1;
        for (auto &s : sizes) s = (uint16_t)(rnd() % (stride + 50));    // larger than the slot too
        orc_depayload(pk.data(), sizes.data(), radio.data(), C, F, stride, n, pay.data(), len.data(), info.data());
        orc_depayload(pk.data(), nullptr, radio.data(), C, F, stride, n, pay.data(), len.data(), info.data());
    }
    for (int mode = 1; mode <= 4; ++mode)
        for (size_t n : {0u, 1u, 2u, 3u, 5u, 7u, 60u, 61u, 100u}) {
            std::vector<uint8_t> in(n + 1), out(n + 1);
            orc_gen_uniform(in.data(), n, rnd(), 0);
            orc_g726_reorder(in.data(), out.data(), n, mode);
        }
    uint8_t hdr[44], ex[320], fr[160];
    orc_wav_header(hdr, 8000, 0xFFFFFFF0u);
    orc_gen_uniform(fr, 160, 1, 0);
    orc_wav_expand(fr, 160, ex);
    orc_ptt_logger l; orc_ptt_init(&l);
    orc_ptt_pressed(&l, 1e9);
    for (int i = 0; i < 70000; ++i) { l.OutgoingRTP = (uint8_t)rnd(); orc_ptt_keeplog(&l, (double)(rnd() % 40000) - 5000.0); }
    orc_ptt_released(&l);
    std::vector<uint8_t> sp((size_t)4 * 3 * 160), cd(4, 0);
    cd[1] = 8;
    orc_gen_speech(sp.data(), cd.data(), 4, 3, 160, 5, 65530, 0);
    printf("oracle pass ok (%u)\n", acc);
}

static void host_pass(const char *tmpdir)
{
    // GPU-free entry points of the mirror: packet parse hooks without an instance, recorder, FIFO error paths
    uint8_t pkt[400];
    for (int radio = 0; radio < 2; ++radio) {
        tp_adapter *a = igdsp_host_adapter_new(5 + radio, radio);
        for (int i = 0; i < 4000; ++i) {
            const long size = (long)(rnd() % 330);
            orc_gen_uniform(pkt, sizeof pkt, rnd(), 0);
            if (rnd() & 1) { pkt[0] = 0x90; pkt[1] = (uint8_t)((rnd() & 1) ? 8 : (rnd() & 1) ? 0 : 123); }
            transport_rtp_cb(a, pkt, size);
            transport_send_rtp(a, pkt, (size_t)size);
            custom_rtp_hdr *h = nullptr;
            if (size >= 20) decodeRtp(pkt, &h);
        }
        transport_rtp_cb(a, pkt, 0);
        transport_rtp_cb(a, pkt, -1);
        igdsp_host_adapter_free(a);
    }
    char path[512];
    snprintf(path, sizeof path, "%s/san.wav", tmpdir);
    void *w = igdsp_wav_start(path, 8000);
    if (!w) { fprintf(stderr, "wav_start failed\n"); exit(2); }
    for (int i = 0; i < 50; ++i) {
        orc_gen_uniform(pkt, 180, rnd(), 0);
        igdsp_wav_writeRTPWav(w, (const char *)pkt, (const char *)pkt + 20, 180, (unsigned)(rnd() % 161));
    }
    igdsp_wav_writeRTPWav(w, (const char *)pkt, (const char *)pkt + 20, 20, 0);
    igdsp_wav_stop(w);
    if (igdsp_wav_start("/nonexistent-dir/x.wav", 8000) != nullptr) { fprintf(stderr, "wav_start should fail\n"); exit(2); }
    if (igdsp_meter_fifo_open("san-no-reader", 10) >= 0) { fprintf(stderr, "fifo_open should time out\n"); exit(2); }
    igdsp_meter_fifo_write(-1, 5);
    igdsp_meter_fifo_close(-1);
    // no device in this container: creation must fail cleanly and every handle entry must tolerate NULL
    void *h = igdsp_host_create(0, 8);
    if (h) {                                   // (a GPU box) exercise the PTT logger text with tiny buffers
        char js[8];
        igdsp_host_bind_radio(h, 0, 5);
        igdsp_host_keeplog(h, 0, 123.0);
        igdsp_host_ptt_event(h, 0, "PTT", "sip:x", 1.0, js, sizeof js);
        igdsp_host_ptt_event(h, 0, "PTT", "sip:x", 1.0, js, 0);
        igdsp_host_destroy(h);
    } else {
        uint32_t n = 0; trx t; ptt_window pw; char js[64];
        igdsp_host_tick(nullptr, &n); igdsp_host_get_trx(nullptr, 0, &t); igdsp_host_bind_radio(nullptr, 0, 1);
        igdsp_host_keeplog(nullptr, 0, 1.0); igdsp_host_ptt_event(nullptr, 0, "a", "b", 1.0, js, sizeof js);
        igdsp_host_get_window(nullptr, 0, &pw); igdsp_host_set_mode(nullptr, SERVER, 0); igdsp_host_ed137_events(nullptr);
        igdsp_host_destroy(nullptr);
    }
    printf("host pass ok\n");
}

int main(int argc, char **argv)
{
    oracle_pass();
    host_pass(argc > 1 ? argv[1] : ".");
    return 0;
}
