"""CPU checks of the C++ host mirror: the WavWriter-compatible recorder reproduces the REAL
reference recorder's bytes (golden fixture written by /root/reference/WavWriter.cpp), and the
library loads without a GPU."""
import hashlib
import json
import os

import numpy as np

from tests import host_util as hu


def test_recorder_bytes_equal_real_wavwriter(golden_dir, tmp_path):
    L = hu.load()
    g = np.load(os.path.join(golden_dir, "config1_4ch_50f.npz"))
    with open(os.path.join(golden_dir, "config1_4ch_50f.json")) as fh:
        meta = json.load(fh)
    pkt = bytes(12)
    for c in range(4):
        path = str(tmp_path / f"rec{c}.wav").encode()
        w = L.igdsp_wav_start(path, 8000)
        assert w
        for f in range(50):
            pl = g["payload"][f, c].tobytes()
            assert L.igdsp_wav_writeRTPWav(w, pkt, pl, 12, len(pl)) == 0
        assert L.igdsp_wav_stop(w) == 0
        mine = open(path, "rb").read()
        assert len(mine) == meta["wav_len"][c]
        assert hashlib.sha256(mine).hexdigest() == meta["wav_sha256"][c]
        assert mine == g[f"wav{c}"].tobytes()


def test_recorder_rejects_bad_args(tmp_path):
    L = hu.load()
    assert L.igdsp_wav_writeRTPWav(None, None, None, 0, 0) == -22
    assert not L.igdsp_wav_start(str(tmp_path / "nodir" / "x.wav").encode(), 8000)
    assert L.igdsp_wav_stop(None) == -22


def test_host_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present")
    L = hu.load()
    assert not L.igdsp_host_create(0, 4)          # NULL: no device, no CPU metering path
    assert L.igdsp_host_tick(None, None) == -22


def test_transport_rtp_cb_fuzz_against_depayload_oracle(orc):
    """CPU only: the host mirror's transport_rtp_cb (header parse, PT gate, payload copy, keep-alive / oversize
    handling) against the oracle's restatement of TransportAdapter.cpp:240-292 on random packets.  No RoIP_ED137
    instance exists here, so nothing is metered — this exercises the parsing and the 256-byte buffer guard."""
    import socket

    L = hu.load()
    rng = np.random.default_rng(11)
    adapters = {0: L.igdsp_host_adapter_new(1, 0), 1: L.igdsp_host_adapter_new(2, 1)}
    for it in range(400):
        radio = int(rng.integers(0, 2))
        hdr = 20 if radio else 12
        pt = int(rng.choice([0, 8, 18, 123, 96, 3]))
        plen = int(rng.choice([160, 160, 24, 164, 0, 255, 256, 257, 300, 1]))
        size = hdr + plen
        pkt = bytearray(hu.rtp_packet(pt, it, rng.integers(0, 256, plen, dtype=np.uint8).tobytes(), bool(radio), int(rng.integers(0, 2 ** 32))))
        if it % 9 == 0:
            size = int(rng.integers(0, hdr))                     # runt
        a = adapters[radio].contents
        before = (a.payload_bufSize, bytes(a.payload_buff))
        L.transport_rtp_cb(adapters[radio], bytes(pkt), size)
        stride = 320
        slot = np.zeros((1, 1, stride), np.uint8)
        slot[0, 0, :len(pkt)] = np.frombuffer(bytes(pkt), np.uint8)
        _, _, info = orc.depayload(slot, np.array([[size]], np.uint16), [radio], n=256)
        fl = int(info["flags"][0, 0])
        if size < 12 or (fl & 0x40) or (fl & 0x80):              # runt / oversize: buffers untouched
            assert (a.payload_bufSize, bytes(a.payload_buff)) == before
            continue
        assert a.payload_bufSize == int(info["payload_len"][0, 0]) == size - hdr
        assert bytes(a.payload_buff[: size - hdr]) == bytes(pkt[hdr:size])
        assert a.last_rx_pt == int(info["pt"][0, 0])
        if radio and pt in (0, 8, 18, 123):
            assert socket.ntohl(a.ed137_value) == int(info["ed137"][0, 0])
        assert a.rtpAudio == (0 if pt == 123 else 1)
    # TX side: silence probe counter and size guards, still no instance
    sil = hu.rtp_packet(8, 1, bytes([0xD5]) * 160, radio=False)
    for i in range(3):
        assert L.transport_send_rtp(adapters[0], sil, len(sil)) == 0 and adapters[0].contents.rtpFalse == i + 1
    assert L.transport_send_rtp(adapters[0], hu.rtp_packet(8, 2, bytes(160), radio=False), 172) == 0
    assert adapters[0].contents.rtpFalse == 0
    assert L.transport_send_rtp(adapters[0], sil, 300) == -22 and L.transport_send_rtp(adapters[0], sil, 5) == -22
    for a in adapters.values():
        L.igdsp_host_adapter_free(a)


def test_fifo_writer_drives_the_real_reference_audiometer(orc):
    """The product's FIFO producer (igdsp_meter_fifo_*) feeding the REAL reference AudioMeter::getAudioLevel()
    (oracle/_ref, built from audiometer.cpp + moc): the reference's own consumer code emits exactly
    int(float(v*100.0/30000.0)) for every level we write."""
    import ctypes as C
    import threading
    import time

    import pytest

    if not orc.ref_audiometer_available():
        pytest.skip("oracle/_ref/libref_audiometer.so not built (reference / Qt not present)")
    L = hu.load()
    R = C.CDLL(os.path.join(os.path.dirname(orc.__file__), "_ref", "libref_audiometer.so"))
    R.ref_audiometer_consume.restype = C.c_int
    card = f"igdspw{os.getpid()}".encode()
    levels = [0, 150, 300, 10138, 29999, 30000, 32124, 32256]
    assert L.igdsp_meter_fifo_open(b"nobody-listens", 5) == -2          # no reader: IGDSP_ENOENT after the timeout

    def produce():
        fd = L.igdsp_meter_fifo_open(card, 5000)
        assert fd >= 0
        for v in levels:
            assert L.igdsp_meter_fifo_write(fd, v) == 0
            time.sleep(0.002)
        assert L.igdsp_meter_fifo_close(fd) == 0

    t = threading.Thread(target=produce)
    t.start()
    out = (C.c_int * 64)()
    n = R.ref_audiometer_consume(card, 64, out)
    t.join()
    os.unlink(f"/tmp/capturefifo{card.decode()}")
    assert n == len(levels)
    assert list(out[:n]) == [orc.percent(v) for v in levels]
