"""-m gpu: SURVEY 8(f) rank 1 — ED-137 RTP depayload + gather on the device, against the oracle's
restatement of transport_rtp_cb (TransportAdapter.cpp:240-292) AND against the C++ host mirror's
own transport_rtp_cb fed the same packets (two independent implementations)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from igate4xsoftphonedsp_amd import capi  # noqa: E402
from tests import gpu_util as gu  # noqa: E402
from tests import host_util as hu  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(device=0, max_channels=64)
    yield c
    c.close()


def _make_packets(orc, C_, F_, stride, n, seed=1):
    rng = np.random.default_rng(seed)
    radio = (np.arange(C_) % 3 != 0).astype(np.uint8)
    pk = orc.gen_uniform(F_ * C_ * stride, seed=seed).reshape(F_, C_, stride).copy()   # garbage beyond each packet
    sizes = np.zeros((F_, C_), np.uint16)
    for f in range(F_):
        for c in range(C_):
            hdr = 20 if radio[c] else 12
            kind = rng.integers(0, 12)
            pt = [0, 8, 0, 8, 0, 8, 123, 18, 96, 0, 8, 0][kind]
            plen = [n, n, n, n, 24, n - 1, 0, 20, n, n + 4, 0, 1][kind]
            size = hdr + plen
            if kind == 11 and rng.integers(0, 2):
                size = int(rng.integers(0, hdr))                          # runt
            size = min(size, stride)
            pkt = bytearray(hu.rtp_packet(pt, f, bytes(pk[f, c, hdr:hdr + max(plen, 0)]), bool(radio[c]), int(rng.integers(0, 2 ** 32))))
            if rng.integers(0, 5) == 0:
                pkt[0] &= 0x3F                                            # wrong version
            if radio[c] and rng.integers(0, 5) == 0:
                pkt[13] = 0x66                                            # wrong extension profile
            if rng.integers(0, 4) == 0:
                pkt[1] |= 0x80                                            # marker
            pkt = bytes(pkt)[:stride]
            pk[f, c, :len(pkt)] = np.frombuffer(pkt, np.uint8)
            sizes[f, c] = size
    return pk, sizes, radio


@pytest.mark.parametrize("stride,n", [(180, 160), (192, 160), (276, 256), (64, 20)])
def test_depayload_vs_oracle(ctx, orc, stride, n):
    torch = gu.torch_cuda()
    C_, F_ = 29, 11
    pk, sizes, radio = _make_packets(orc, C_, F_, stride, n, seed=stride + n)
    d_pl, d_len, d_info = gu.dev_zeros(F_ * C_ * n, 0xEE), gu.dev_zeros(F_ * C_ * 2, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE)
    ctx.depayload(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(radio), C_, F_, stride, n, d_pl, d_len, d_info)
    torch.cuda.synchronize()
    epl, elen, einfo = orc.depayload(pk, sizes, radio, n)
    assert np.array_equal(gu.to_host(d_len, "<u2", (F_, C_)), elen)
    ginfo = gu.to_host(d_info, capi.RTP_INFO, (F_, C_))
    for f in capi.RTP_INFO.names:
        assert np.array_equal(ginfo[f], einfo[f]), f
    assert np.array_equal(gu.to_host(d_pl, np.uint8, (F_, C_, n)), epl)
    # every class of packet occurred
    fl = einfo["flags"]
    for bit in (capi.RTP_KEEPALIVE, capi.RTP_METERED, capi.RTP_RUNT, capi.RTP_OVERSIZE, capi.RTP_ED137_OK):
        assert (fl & bit).any(), bit
    # ED-137 word fields with the reference's masks (Functions.cpp:1018,1045,1136,1148)
    v = einfo["ed137"].astype(np.uint64)
    assert np.array_equal((v & 0xe0000000) >> 29, (ginfo["ed137"].astype(np.uint64) >> 29) & 7)


@pytest.mark.parametrize("stride,C_,F_,with_sizes", [(180, 32, 6, True), (184, 96, 4, True), (256, 64, 3, True), (180, 64, 2, False), (192, 192, 1, False)])
def test_depayload_tuned_path_vs_oracle(ctx, orc, stride, C_, F_, with_sizes):
    """n == 160 and C*F % 64 == 0 take k_depayload64 (header parsed once per packet, pieces spread like the fused kernel):
    mixed 12/20-byte headers, runts, oversize, keep-alives, partial lengths, garbage behind short packets."""
    torch = gu.torch_cuda()
    n = 160
    pk, sizes, radio = _make_packets(orc, C_, F_, stride, n, seed=stride * 7 + C_)
    d_pl, d_len, d_info = gu.dev_zeros(F_ * C_ * n, 0xEE), gu.dev_zeros(F_ * C_ * 2, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE)
    ctx.depayload(gu.to_dev(pk), gu.to_dev(sizes) if with_sizes else None, gu.to_dev(radio), C_, F_, stride, n, d_pl, d_len, d_info)
    torch.cuda.synchronize()
    epl, elen, einfo = orc.depayload(pk, sizes if with_sizes else None, radio, n)
    assert np.array_equal(gu.to_host(d_len, "<u2", (F_, C_)), elen)
    ginfo = gu.to_host(d_info, capi.RTP_INFO, (F_, C_))
    for f in capi.RTP_INFO.names:
        assert np.array_equal(ginfo[f], einfo[f]), f
    assert np.array_equal(gu.to_host(d_pl, np.uint8, (F_, C_, n)), epl)
    if with_sizes:
        assert ((elen > 0) & (elen < n)).any() and (elen == 0).any() and (elen == n).any()


def test_depayload_then_meter_equals_direct(ctx, orc):
    """packets -> igdsp_depayload -> igdsp_decode_meter(len) == oracle meter on the oracle's payloads;
    full 180-byte slots without a size array take the same route."""
    torch = gu.torch_cuda()
    C_, F_, stride, n = 64, 6, 180, 160
    pk, sizes, radio = _make_packets(orc, C_, F_, stride, n, seed=77)
    codec = np.where(np.arange(C_) & 1, 8, 0).astype(np.uint8)
    d_pl, d_len, d_info = gu.dev_zeros(F_ * C_ * n), gu.dev_zeros(F_ * C_ * 2), gu.dev_zeros(F_ * C_ * 8)
    d_st = gu.dev_zeros(F_ * C_ * 16, 0xEE)
    d_cd = gu.to_dev(codec)
    ctx.depayload(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(radio), C_, F_, stride, n, d_pl, d_len, d_info)
    ctx.decode_meter(d_pl, d_cd, C_, F_, n, d_st, length=d_len)
    torch.cuda.synchronize()
    epl, elen, _ = orc.depayload(pk, sizes, radio, n)
    est = orc.decode_meter(epl, codec, length=elen)
    gu.assert_stats_equal(gu.to_host(d_st, capi.FRAME_STATS, (F_, C_)), est, n=elen)
    # sizes == NULL: every slot is a full radio packet (the steady state) -> dense payload for the fast path
    radio1 = np.ones((C_,), np.uint8)
    body = orc.gen_uniform(F_ * C_ * n, seed=3).reshape(F_, C_, n)
    pk2 = np.zeros((F_, C_, stride), np.uint8)
    for f in range(F_):
        for c in range(C_):
            pk2[f, c] = np.frombuffer(hu.rtp_packet(int(codec[c]), f, body[f, c].tobytes(), True, 0x20000000), np.uint8)
    ctx.depayload(gu.to_dev(pk2), None, gu.to_dev(radio1), C_, F_, stride, n, d_pl, d_len, d_info)
    ctx.decode_meter(d_pl, d_cd, C_, F_, n, d_st)
    torch.cuda.synchronize()
    assert np.array_equal(gu.to_host(d_pl, np.uint8, (F_, C_, n)), body)
    assert np.all(gu.to_host(d_len, "<u2") == n)
    gu.assert_stats_equal(gu.to_host(d_st, capi.FRAME_STATS, (F_, C_)), orc.decode_meter(body, codec), n=n)


def test_depayload_agrees_with_host_transport_rtp_cb(ctx, orc):
    """The C++ host mirror's transport_rtp_cb (an independent implementation of TransportAdapter.cpp:240-292)
    sees the same payload bytes / lengths / ED-137 words as the device kernel."""
    torch = gu.torch_cuda()
    L = hu.load()
    C_, F_, stride, n = 6, 9, 276, 256
    pk, sizes, radio = _make_packets(orc, C_, F_, stride, n, seed=5)
    d_pl, d_len, d_info = gu.dev_zeros(F_ * C_ * n), gu.dev_zeros(F_ * C_ * 2), gu.dev_zeros(F_ * C_ * 8)
    ctx.depayload(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(radio), C_, F_, stride, n, d_pl, d_len, d_info)
    torch.cuda.synchronize()
    gpl, ginfo = gu.to_host(d_pl, np.uint8, (F_, C_, n)), gu.to_host(d_info, capi.RTP_INFO, (F_, C_))
    adapters = [L.igdsp_host_adapter_new(100 + c, int(radio[c])) for c in range(C_)]
    checked = 0
    for f in range(F_):
        for c in range(C_):
            size = int(sizes[f, c])
            hdr = 20 if radio[c] else 12
            if size < hdr:
                continue                                                    # the mirror ignores runts
            a = adapters[c].contents
            before = a.payload_bufSize
            L.transport_rtp_cb(adapters[c], pk[f, c].tobytes(), size)
            if size - hdr > 256:
                assert a.payload_bufSize == before
                continue
            assert a.payload_bufSize == int(ginfo[f, c]["payload_len"]) == size - hdr
            if ginfo[f, c]["flags"] & capi.RTP_METERED:
                assert bytes(a.payload_buff[: size - hdr]) == gpl[f, c, : size - hdr].tobytes()
                checked += 1
            if radio[c] and int(ginfo[f, c]["pt"]) in (0, 8, 18, 123):
                import socket
                assert socket.ntohl(a.ed137_value) == int(ginfo[f, c]["ed137"])
    assert checked > 10
    for a in adapters:
        L.igdsp_host_adapter_free(a)


def _to_slots(pk180, sizes):
    """[F][C][180] packets + sizes -> [F][C][192] slots: u16 size, 10 reserved bytes, packet at +12."""
    F_, C_, _ = pk180.shape
    slots = np.zeros((F_, C_, 192), np.uint8)
    slots[:, :, 0] = sizes & 0xFF
    slots[:, :, 1] = sizes >> 8
    slots[:, :, 12:] = pk180
    return slots


@pytest.mark.parametrize("C_,F_", [(64, 3), (96, 2), (256, 5), (32, 2)])
def test_fused_rtp_slots_meter(ctx, orc, C_, F_):
    """igdsp_decode_meter_rtp: one kernel from packet slots to records.  info == the depayload oracle;
    metered frames (20 < size <= 180, PT == codec; short payloads over their own length) == the meter oracle on the
    depayload oracle's (payload, len); everything else EMPTY; aggregate over metered only."""
    torch = gu.torch_cuda()
    n = 160
    pk, sizes, _ = _make_packets(orc, C_, F_, 180, n, seed=C_ * 10 + F_)
    radio = np.ones((C_,), np.uint8)
    # re-make as all-radio packets with mostly full audio frames
    rng = np.random.default_rng(C_ + F_)
    codec = np.where(np.arange(C_) % 3 == 0, 8, 0).astype(np.uint8)
    for f in range(F_):
        for c in range(C_):
            kind = rng.integers(0, 10)
            pt = int(codec[c]) if kind < 6 else [123, 18, 8 - int(codec[c]), 96][kind - 6]
            plen = n if kind != 9 else 24
            if kind == 6:
                plen = 0
            if kind in (4, 5):
                plen = int(rng.choice([1, 3, 24, 48, 49, 52, 80, 157, 159]))       # PT matches, 0 < payloadlen < 160: metered over plen bytes
            body = orc.gen_uniform(max(plen, 1), seed=f * 1000 + c).tobytes()[:plen]
            pkt = hu.rtp_packet(pt, f, body, True, int(rng.integers(0, 2 ** 32)))
            pk[f, c, :] = orc.gen_uniform(180, seed=7).astype(np.uint8)       # stale bytes behind short packets
            pk[f, c, :len(pkt)] = np.frombuffer(pkt, np.uint8)
            sizes[f, c] = len(pkt) if kind != 8 or f % 2 else int(rng.integers(0, 20))   # some runts
    slots = _to_slots(pk, sizes)
    d_st, d_info, d_agg = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
    ctx.decode_meter_rtp(gu.to_dev(slots), gu.to_dev(codec), C_, F_, d_st, info=d_info, agg=d_agg, rank=2)
    torch.cuda.synchronize()
    epl, elen, einfo = orc.depayload(pk, sizes, radio, n)
    ginfo = gu.to_host(d_info, capi.RTP_INFO, (F_, C_))
    for fld in capi.RTP_INFO.names:
        assert np.array_equal(ginfo[fld], einfo[fld]), fld
    metered = (sizes > 20) & (sizes <= 180) & (einfo["pt"] == codec[None, :]) & np.isin(einfo["pt"], (0, 8))
    assert metered.any() and (~metered).any() and (metered & (sizes < 180)).any()
    est = orc.decode_meter(epl, codec, length=elen)       # short payloads are metered over their own length (TransportAdapter.cpp:270-291)
    gst = gu.to_host(d_st, capi.FRAME_STATS, (F_, C_))
    gu.assert_stats_equal(gst[metered].reshape(1, -1), est[metered].reshape(1, -1), n=elen[metered].reshape(1, -1))
    emp = gst[~metered]
    assert np.all(emp["flags"] == capi.FLAG_EMPTY) and np.all(emp["sumsq"] == 0) and np.all(emp["peak"] == 0) and np.all(emp["rms"] == 0)
    agg = gu.to_host(d_agg, capi.AGGREGATE)[0]
    assert int(agg["frames"]) == int(metered.sum()) and int(agg["samples"]) == int(elen[metered].sum())
    assert int(agg["byte_mean_sum"]) == int(est["byte_mean"][metered].sum()) and int(agg["n_silent"]) == int(((est["flags"] & capi.FLAG_SILENT) != 0)[metered].sum())
    assert int(agg["sumsq"]) == int(est["sumsq"][metered].sum(dtype=np.uint64))
    assert int(agg["peak_slot"][2]) == int(est["peak"][metered].max())
    # shape / alignment rules are reported, not silently re-routed
    assert ctx.L.igdsp_decode_meter_rtp(ctx.h, d_st.data_ptr(), d_st.data_ptr(), 33, 1, d_st.data_ptr(), None, None, 0, None) == -22


@pytest.mark.parametrize("hdr,stride,C_,F_,with_sizes", [(20, 180, 64, 3, True), (20, 184, 96, 2, True), (12, 172, 64, 2, True),
                                                         (12, 180, 32, 4, True), (20, 256, 128, 3, True), (20, 180, 64, 2, False)])
def test_fused_packed_packets_meter(ctx, orc, hdr, stride, C_, F_, with_sizes):
    """igdsp_decode_meter_packets: the same fused kernel over packets packed at their natural stride (dword-aligned
    loads), one header type per launch.  Same oracle relations as the slot form."""
    torch = gu.torch_cuda()
    n = 160
    radio = np.full((C_,), 1 if hdr == 20 else 0, np.uint8)
    rng = np.random.default_rng(hdr * 1000 + stride + C_)
    codec = np.where(np.arange(C_) % 3 == 0, 8, 0).astype(np.uint8)
    pk = orc.gen_uniform(F_ * C_ * stride, seed=stride).reshape(F_, C_, stride).copy()
    sizes = np.zeros((F_, C_), np.uint16)
    for f in range(F_):
        for c in range(C_):
            kind = int(rng.integers(0, 10)) if with_sizes else int(rng.integers(0, 6))
            pt = int(codec[c]) if kind < 6 else [123, 18, 8 - int(codec[c]), 96][kind - 6]
            plen = n if kind != 9 else 24
            if kind == 6:
                plen = 0
            if kind in (4, 5):
                plen = int(rng.choice([1, 3, 24, 48, 49, 52, 80, 157, 159]))       # PT matches, 0 < payloadlen < 160: metered over plen bytes
            if not with_sizes and rng.integers(0, 4) == 0:
                pt = 8 - int(codec[c])                                       # PT mismatch at full size
            body = orc.gen_uniform(max(plen, 1), seed=f * 1000 + c).tobytes()[:plen]
            pkt = bytearray(hu.rtp_packet(pt, f, body, hdr == 20, int(rng.integers(0, 2 ** 32))))
            if rng.integers(0, 6) == 0:
                pkt[1] |= 0x80
            if hdr == 20 and rng.integers(0, 6) == 0:
                pkt[13] = 0x66
            pk[f, c, :len(pkt)] = np.frombuffer(bytes(pkt), np.uint8)
            sizes[f, c] = len(pkt) if kind != 8 or f % 2 else int(rng.integers(0, hdr))
    d_st, d_info, d_agg = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
    ctx.decode_meter_packets(gu.to_dev(pk), gu.to_dev(sizes) if with_sizes else None, gu.to_dev(codec), C_, F_, stride, hdr,
                             d_st, info=d_info, agg=d_agg, rank=1)
    torch.cuda.synchronize()
    epl, elen, einfo = orc.depayload(pk, sizes if with_sizes else None, radio, n)
    ginfo = gu.to_host(d_info, capi.RTP_INFO, (F_, C_))
    for fld in capi.RTP_INFO.names:
        assert np.array_equal(ginfo[fld], einfo[fld]), fld
    full = ((sizes > hdr) & (sizes <= hdr + n)) if with_sizes else np.ones((F_, C_), bool)
    metered = full & (einfo["pt"] == codec[None, :]) & np.isin(einfo["pt"], (0, 8))
    assert metered.any() and (~metered).any()
    if with_sizes:
        assert (metered & (sizes < hdr + n)).any()
    est = orc.decode_meter(epl, codec, length=elen)
    gst = gu.to_host(d_st, capi.FRAME_STATS, (F_, C_))
    gu.assert_stats_equal(gst[metered].reshape(1, -1), est[metered].reshape(1, -1), n=elen[metered].reshape(1, -1))
    emp = gst[~metered]
    assert np.all(emp["flags"] == capi.FLAG_EMPTY) and np.all(emp["sumsq"] == 0) and np.all(emp["peak"] == 0)
    agg = gu.to_host(d_agg, capi.AGGREGATE)[0]
    assert int(agg["frames"]) == int(metered.sum()) and int(agg["samples"]) == int(elen[metered].sum())
    assert int(agg["sumsq"]) == int(est["sumsq"][metered].sum(dtype=np.uint64))
    assert int(agg["peak_slot"][1]) == int(est["peak"][metered].max())
    # packed and slot forms agree record for record on 180-byte radio packets
    if hdr == 20 and stride == 180 and with_sizes:
        d_st2 = gu.dev_zeros(F_ * C_ * 16, 0xEE)
        ctx.decode_meter_rtp(gu.to_dev(_to_slots(pk, sizes)), gu.to_dev(codec), C_, F_, d_st2)
        torch.cuda.synchronize()
        assert gu.to_host(d_st2, np.uint8).tobytes() == gu.to_host(d_st, np.uint8).tobytes()
    # argument rules
    L, p = ctx.L, d_st.data_ptr()
    assert L.igdsp_decode_meter_packets(ctx.h, p, None, p, 64, 1, 180, 16, p, None, None, 0, None) == -22   # header type
    assert L.igdsp_decode_meter_packets(ctx.h, p, None, p, 64, 1, 178, 20, p, None, None, 0, None) == -22   # stride < hdr+160
    assert L.igdsp_decode_meter_packets(ctx.h, p, None, p, 64, 1, 182, 20, p, None, None, 0, None) == -22   # stride % 4
    assert L.igdsp_decode_meter_packets(ctx.h, p, None, p, 33, 1, 180, 20, p, None, None, 0, None) == -22   # C*F % 64


@pytest.mark.parametrize("stride,C_,F_", [(180, 64, 3), (184, 96, 2), (256, 128, 2)])
def test_fused_packed_packets_mixed_headers(ctx, orc, stride, C_, F_):
    """igdsp_decode_meter_packets_mixed: radio (20-byte header) and SIP (12-byte header) channels in ONE launch; same
    oracle relations as the single-header form, per channel."""
    torch = gu.torch_cuda()
    n = 160
    rng = np.random.default_rng(stride + C_)
    radio = (rng.integers(0, 2, C_)).astype(np.uint8)
    radio[:4] = [1, 0, 0, 1]
    codec = np.where(np.arange(C_) % 3 == 0, 8, 0).astype(np.uint8)
    pk = orc.gen_uniform(F_ * C_ * stride, seed=stride).reshape(F_, C_, stride).copy()
    sizes = np.zeros((F_, C_), np.uint16)
    for f in range(F_):
        for c in range(C_):
            hdr = 20 if radio[c] else 12
            kind = int(rng.integers(0, 10))
            pt = int(codec[c]) if kind < 6 else [123, 18, 8 - int(codec[c]), 96][kind - 6]
            plen = n if kind != 9 else 24
            if kind == 6:
                plen = 0
            if kind in (4, 5):
                plen = int(rng.choice([1, 3, 24, 48, 49, 52, 80, 157, 159]))       # PT matches, 0 < payloadlen < 160: metered over plen bytes
            body = orc.gen_uniform(max(plen, 1), seed=f * 1000 + c).tobytes()[:plen]
            pkt = bytearray(hu.rtp_packet(pt, f, body, bool(radio[c]), int(rng.integers(0, 2 ** 32))))
            if rng.integers(0, 6) == 0:
                pkt[1] |= 0x80
            pk[f, c, :len(pkt)] = np.frombuffer(bytes(pkt), np.uint8)
            sizes[f, c] = len(pkt) if kind != 8 or f % 2 else int(rng.integers(0, hdr))
    d_st, d_info, d_agg = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
    ctx.decode_meter_packets_mixed(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), gu.to_dev(radio), C_, F_, stride, d_st, info=d_info, agg=d_agg, rank=3)
    torch.cuda.synchronize()
    epl, elen, einfo = orc.depayload(pk, sizes, radio, n)
    ginfo = gu.to_host(d_info, capi.RTP_INFO, (F_, C_))
    for fld in capi.RTP_INFO.names:
        assert np.array_equal(ginfo[fld], einfo[fld]), fld
    hdrs = np.where(radio, 20, 12)[None, :]
    metered = (sizes > hdrs) & (sizes <= hdrs + n) & (einfo["pt"] == codec[None, :]) & np.isin(einfo["pt"], (0, 8))
    assert metered[:, radio == 1].any() and metered[:, radio == 0].any() and (~metered).any()
    short = metered & (sizes < hdrs + n)
    assert short[:, radio == 1].any() and short[:, radio == 0].any()
    est = orc.decode_meter(epl, codec, length=elen)
    gst = gu.to_host(d_st, capi.FRAME_STATS, (F_, C_))
    gu.assert_stats_equal(gst[metered].reshape(1, -1), est[metered].reshape(1, -1), n=elen[metered].reshape(1, -1))
    emp = gst[~metered]
    assert np.all(emp["flags"] == capi.FLAG_EMPTY) and np.all(emp["sumsq"] == 0) and np.all(emp["peak"] == 0)
    agg = gu.to_host(d_agg, capi.AGGREGATE)[0]
    assert int(agg["frames"]) == int(metered.sum()) and int(agg["samples"]) == int(elen[metered].sum())
    assert int(agg["sumsq"]) == int(est["sumsq"][metered].sum(dtype=np.uint64))
    assert int(agg["peak_slot"][3]) == int(est["peak"][metered].max())
    # all-radio through the mixed entry == the single-header entry, record for record
    ones = np.ones((C_,), np.uint8)
    d_a, d_b = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 16, 0xEE)
    ctx.decode_meter_packets_mixed(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), gu.to_dev(ones), C_, F_, stride, d_a)
    ctx.decode_meter_packets(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), C_, F_, stride, 20, d_b)
    torch.cuda.synchronize()
    assert gu.to_host(d_a, np.uint8).tobytes() == gu.to_host(d_b, np.uint8).tobytes()
    p_ = d_st.data_ptr()
    assert ctx.L.igdsp_decode_meter_packets_mixed(ctx.h, p_, None, p_, None, 64, 1, 180, p_, None, None, 0, None) == -22   # radio missing
    assert ctx.L.igdsp_decode_meter_packets_mixed(ctx.h, p_, None, p_, p_, 64, 1, 176, p_, None, None, 0, None) == -22    # stride < 180


def test_fuzz_fused_packets_against_two_step_route(ctx, orc):
    """40 random cases of the fused packet entries (slot form, packed form with 12- / 20-byte headers and strides 172..256,
    mixed headers): random sizes incl. runts, short payloads of every length, keep-alives, foreign PTs, PT / codec mismatch.
    Each must equal the oracle's depayload followed by the oracle's meter over (payload, len)."""
    torch = gu.torch_cuda()
    rng = np.random.default_rng(4242)
    n = 160
    for case in range(40):
        form = int(rng.integers(0, 3))                                  # 0 slots, 1 packed single header, 2 packed mixed
        C_ = int(rng.choice([64, 128, 192, 320]))
        F_ = int(rng.integers(1, 4))
        hdr = 20 if form == 0 else int(rng.choice([12, 20]))
        stride = 180 if form == 0 else int(rng.choice([hdr + 160, 180, 184, 200, 256]))
        stride = max(stride, 180 if form == 2 else hdr + 160)
        radio = np.ones((C_,), np.uint8) if form == 0 else (np.full((C_,), hdr == 20, np.uint8) if form == 1 else rng.integers(0, 2, C_).astype(np.uint8))
        codec = rng.choice(np.array([0, 8], np.uint8), size=C_)
        pk = orc.gen_uniform(F_ * C_ * stride, seed=900 + case).reshape(F_, C_, stride).copy()
        sizes = np.zeros((F_, C_), np.uint16)
        for f in range(F_):
            for c in range(C_):
                h = 20 if radio[c] else 12
                kind = int(rng.integers(0, 12))
                pt = int(codec[c]) if kind < 8 else [123, 18, 8 - int(codec[c]), 96][kind - 8]
                plen = n if kind < 5 else (int(rng.integers(1, n)) if kind < 8 else int(rng.choice([0, 24, n])))
                pkt = bytearray(hu.rtp_packet(pt, f, bytes(pk[f, c, h:h + plen]), bool(radio[c]), int(rng.integers(0, 2 ** 32))))
                if rng.integers(0, 8) == 0:
                    pkt[1] |= 0x80
                pk[f, c, :len(pkt)] = np.frombuffer(bytes(pkt), np.uint8)
                sizes[f, c] = len(pkt) if rng.integers(0, 15) else int(rng.integers(0, h))
        d_st, d_info, d_agg = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
        if form == 0:
            ctx.decode_meter_rtp(gu.to_dev(_to_slots(pk, sizes)), gu.to_dev(codec), C_, F_, d_st, info=d_info, agg=d_agg, rank=case % 8)
        elif form == 1:
            ctx.decode_meter_packets(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), C_, F_, stride, hdr, d_st, info=d_info, agg=d_agg, rank=case % 8)
        else:
            ctx.decode_meter_packets_mixed(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), gu.to_dev(radio), C_, F_, stride, d_st, info=d_info, agg=d_agg, rank=case % 8)
        torch.cuda.synchronize()
        epl, elen, einfo = orc.depayload(pk, sizes, radio, n)
        ginfo = gu.to_host(d_info, capi.RTP_INFO, (F_, C_))
        for fld in capi.RTP_INFO.names:
            assert np.array_equal(ginfo[fld], einfo[fld]), (case, fld)
        metered = (elen > 0) & (einfo["pt"] == codec[None, :])
        est = orc.decode_meter(epl, codec, length=elen)
        gst = gu.to_host(d_st, capi.FRAME_STATS, (F_, C_))
        if metered.any():
            gu.assert_stats_equal(gst[metered].reshape(1, -1), est[metered].reshape(1, -1), n=elen[metered].reshape(1, -1))
        assert np.all(gst[~metered]["flags"] == capi.FLAG_EMPTY), case
        agg = gu.to_host(d_agg, capi.AGGREGATE)[0]
        assert int(agg["frames"]) == int(metered.sum()) and int(agg["samples"]) == int(elen[metered].sum()), case
        assert int(agg["sumsq"]) == int(est["sumsq"][metered].sum(dtype=np.uint64)), case
        assert int(agg["peak_slot"][case % 8]) == (int(est["peak"][metered].max()) if metered.any() else 0), case
