"""-m gpu: BASELINE configs[0] — 4 ch x 8 kHz mu-law, 20 ms frames — driven through the C++ host
mirror exactly as pjmedia would drive the reference: RTP packets -> transport_rtp_cb ->
RoIP_ED137::setIncomingRTP(tp_adapter*) -> (GPU at the tick) -> trx->IncomingRTP.
trx level slots must equal the restated reference loop bit-for-bit; RMS/peak follow the fixture."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import host_util as hu  # noqa: E402


@pytest.fixture()
def host():
    L = hu.load()
    h = L.igdsp_host_create(0, 8)
    assert h, "igdsp_host_create failed on a GPU box"
    yield L, h
    L.igdsp_host_destroy(h)


def _trx(L, h, slot):
    t = hu.Trx()
    assert L.igdsp_host_get_trx(h, slot, C.byref(t)) == 0
    return t


def test_config1_rx_plumbing_matches_reference_levels(host, orc, golden_dir):
    L, h = host
    g = np.load(os.path.join(golden_dir, "config1_4ch_50f.npz"))
    payload = g["payload"]                      # [50][4][160], mu-law
    call_ids = [11, 12, 13, 14]
    adapters = [L.igdsp_host_adapter_new(cid, 1) for cid in call_ids]      # radio calls: 20-byte ED-137 header
    seen = []
    cb = hu.STREAM_CB(lambda ud, pkt, size: seen.append(size))             # stands for pjmedia's stream callback
    for slot, cid in enumerate(call_ids):
        assert L.igdsp_host_bind_radio(h, slot, cid) == 0
        adapters[slot].contents.stream_rtp_cb = C.cast(cb, C.c_void_p)
    n_done = C.c_uint32()
    for f in range(50):
        for slot, cid in enumerate(call_ids):
            pkt = hu.rtp_packet(0, f, payload[f, slot].tobytes(), radio=True, ed137_word=0x20000000 | slot)
            assert len(pkt) == 180                                          # TransportAdapter.cpp:846 "size = 180"
            L.transport_rtp_cb(adapters[slot], pkt, len(pkt))
            a = adapters[slot].contents
            assert a.payload_bufSize == 160 and bytes(a.payload_buff[:160]) == payload[f, slot].tobytes()
        if f % 7 == 3:                                                       # R2S keep-alive between audio frames
            ka = hu.rtp_packet(123, f, b"", radio=True)
            L.transport_rtp_cb(adapters[0], ka, len(ka))
        assert L.igdsp_host_tick(h, C.byref(n_done)) == 0 and n_done.value == 4
        for slot in range(4):
            t = _trx(L, h, slot)
            assert t.IncomingRTP == int(g["byte_mean"][f, slot]) == orc.byte_mean(payload[f, slot])
            assert t.in_peak == int(g["audioop_peak"][f, slot])
            ref = np.sqrt(float(g["sumsq"][f, slot]) / 160.0)
            assert abs(t.in_rms - ref) <= 1e-5 * ref + 1e-30
            assert t.in_percent == orc.percent(np.float32(t.in_rms))
            assert t.in_peak_hold == int(g["audioop_peak"][: f + 1, slot].max())
    assert len(seen) == 200 and all(s == 180 for s in seen)                  # keep-alives never reach the stream
    # audio->idle->audio edges on adapter 0 raise ED-137 events (TransportAdapter.cpp:304-315)
    assert L.igdsp_host_ed137_events(h) >= 4 + 2 * 7
    for a in adapters:
        L.igdsp_host_adapter_free(a)


def test_tx_path_quirk_and_probe(host, orc):
    L, h = host
    a = L.igdsp_host_adapter_new(21, 1)
    assert L.igdsp_host_bind_radio(h, 2, 21) == 0
    n_done = C.c_uint32()
    pl = orc.gen_uniform(160, seed=5)
    pkt = hu.rtp_packet(8, 1, pl.tobytes(), radio=False)                     # pjmedia hands a 12-byte-header packet, PCMA
    # default: payload proper
    assert L.igdsp_host_set_mode(h, 1, 0) == 0
    assert L.transport_send_rtp(a, pkt, len(pkt)) == 0
    assert L.igdsp_host_tick(h, C.byref(n_done)) == 0 and n_done.value == 1
    t = _trx(L, h, 2)
    est = orc.decode_meter(pl.reshape(1, 1, 160), [8])[0, 0]
    assert t.OutgoingRTP == int(est["byte_mean"]) and t.out_peak == int(est["peak"])
    # reference quirk: the loop runs over the first payloadlen bytes of header+payload (roip_ed137.cpp:6505-6517)
    assert L.igdsp_host_set_mode(h, 1, 1) == 0
    assert L.transport_send_rtp(a, pkt, len(pkt)) == 0
    assert L.igdsp_host_tick(h, C.byref(n_done)) == 0
    t = _trx(L, h, 2)
    assert t.OutgoingRTP == orc.byte_mean(np.frombuffer(pkt[:160], dtype=np.uint8))
    # silence probe: packet bytes 40/50/60 == 0xD5 count up, anything else resets (TransportAdapter.cpp:657-673)
    sil = hu.rtp_packet(8, 2, bytes([0xD5]) * 160, radio=False)
    for i in range(5):
        L.transport_send_rtp(a, sil, len(sil))
        assert a.contents.rtpFalse == i + 1
    assert L.igdsp_host_tick(h, C.byref(n_done)) == 0
    assert L.igdsp_host_set_mode(h, 1, 0) == 0
    L.transport_send_rtp(a, sil, len(sil))
    assert L.igdsp_host_tick(h, C.byref(n_done)) == 0
    assert _trx(L, h, 2).out_flags & 0x03 == 0x03                            # SILENT + PROBE_D5 from the GPU path
    L.transport_send_rtp(a, pkt, len(pkt))
    assert a.contents.rtpFalse == 0
    assert L.igdsp_host_tick(h, C.byref(n_done)) == 0 and n_done.value == 1
    # CLIENT mode: hooks do not meter (roip_ed137.cpp:6509,6555)
    assert L.igdsp_host_set_mode(h, 2, 0) == 0
    L.transport_send_rtp(a, pkt, len(pkt))
    assert L.igdsp_host_tick(h, C.byref(n_done)) == 0 and n_done.value == 0
    assert L.transport_send_rtp(a, pkt, 4) == -22
    L.igdsp_host_adapter_free(a)


def test_plain_sip_call_12_byte_header_and_oversize(host, orc):
    L, h = host
    a = L.igdsp_host_adapter_new(31, 0)                                      # not a radio call: 12-byte header
    assert L.igdsp_host_bind_radio(h, 1, 31) == 0
    n_done = C.c_uint32()
    pl = orc.gen_uniform(164, seed=9)                                        # the reference anticipates 164-byte payloads
    pkt = hu.rtp_packet(0, 1, pl.tobytes(), radio=False)
    L.transport_rtp_cb(a, pkt, len(pkt))
    assert a.contents.payload_bufSize == 164
    assert L.igdsp_host_tick(h, C.byref(n_done)) == 0 and n_done.value == 1
    assert _trx(L, h, 1).IncomingRTP == orc.byte_mean(pl)
    big = hu.rtp_packet(0, 2, bytes(300), radio=False)                       # > payload_buff[256]: dropped, not overflowed
    L.transport_rtp_cb(a, big, len(big))
    assert a.contents.payload_bufSize == 164
    assert L.igdsp_host_tick(h, C.byref(n_done)) == 0 and n_done.value == 0
    L.igdsp_host_adapter_free(a)
