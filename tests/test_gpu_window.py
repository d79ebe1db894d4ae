"""-m gpu: SURVEY 8(f) rank 1, last clause — squelch / PTT gating of the meter from the ED-137 word on the device, and the
consecutive-silence run (adapter->rtpFalse, TransportAdapter.cpp:657-673).

igdsp_window_update and igdsp_decode_meter_window against the oracle's restatement (oracle/igdsp_oracle.h: orc_window_update)
fed by the two-step route: oracle depayload -> oracle meter -> per-frame gate from the ED-137 word -> window fold.  The fused
entry must also leave exactly the records, info and aggregate of the ungated fused entries."""
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


def _to_slots(pk180, sizes):
    F_, C_, _ = pk180.shape
    slots = np.zeros((F_, C_, 192), np.uint8)
    slots[:, :, 0] = sizes & 0xFF
    slots[:, :, 1] = sizes >> 8
    slots[:, :, 12:] = pk180
    return slots


def _make_traffic(orc, rng, C_, F_, stride, radio, codec, seed):
    """Per channel a sequence of talk / digital-silence stretches (A-law 0xD5 or mu-law 0xFF payloads: the probe bytes match
    only for 0xD5), ED-137 words whose PTT / SQU bits come and go, plus the odd keep-alive, short payload, foreign PT and runt."""
    n = 160
    pk = orc.gen_uniform(F_ * C_ * stride, seed=seed).reshape(F_, C_, stride).copy()
    sizes = np.zeros((F_, C_), np.uint16)
    for c in range(C_):
        h = 20 if radio[c] else 12
        silent, ptt, squ = bool(rng.integers(0, 2)), int(rng.integers(0, 4)), int(rng.integers(0, 2))
        for f in range(F_):
            if rng.integers(0, 6) == 0:
                silent = not silent
            if rng.integers(0, 5) == 0:
                ptt = int(rng.integers(0, 8)) if rng.integers(0, 2) else 0
            if rng.integers(0, 5) == 0:
                squ = int(rng.integers(0, 2))
            kind = int(rng.integers(0, 20))
            pt = int(codec[c]) if kind < 17 else [123, 18, 8 - int(codec[c])][kind - 17]
            plen = n
            if kind == 15:
                plen = int(rng.choice([24, 40, 48, 49, 52, 100, 159]))      # short payloads: <= 48 leave the run alone
            if kind == 16:
                plen = 0
            body = bytes([0xD5]) * plen if silent else bytes(pk[f, c, h:h + plen])
            if silent and rng.integers(0, 12) == 0:
                body = bytes([0xFF]) * plen                                   # mu-law silence: peak <= 8 but no probe match
            word = (ptt << 29) | (squ << 28) | int(rng.integers(0, 1 << 22))
            pkt = hu.rtp_packet(pt, f, body, bool(radio[c]), word)
            pk[f, c, :len(pkt)] = np.frombuffer(pkt, np.uint8)
            sizes[f, c] = len(pkt) if rng.integers(0, 40) else int(rng.integers(0, h))
    return pk, sizes


def _expected(orc, pk, sizes, radio, codec, mode, alarm, gate, hold0, probe0):
    """the two-step route on the CPU: depayload -> meter over (payload, len) -> EMPTY where the fused kernels do not meter
    (PT != codec) -> window fold with the per-frame gate"""
    n = 160
    epl, elen, einfo = orc.depayload(pk, sizes, radio, n)
    metered = (elen > 0) & (einfo["pt"] == codec[None, :])
    est = orc.decode_meter(epl, codec, length=elen)
    est[~metered] = np.zeros((), est.dtype)
    est["flags"][~metered] = capi.FLAG_EMPTY
    hold, probe = hold0.copy(), probe0.copy()
    orc.window_update(est, hold, info=einfo, n=n, gate_mode=mode, alarm=alarm, gate=gate, probe=probe)
    return est, einfo, elen, metered, hold, probe


def _start_state(rng, C_):
    hold = gu.new_hold(C_)
    hold["count"] = rng.integers(0, 5, C_)
    hold["level_sum"] = rng.integers(0, 500, C_)
    hold["sumsq_acc"] = rng.integers(0, 1 << 40, C_, dtype=np.uint64)
    hold["peak_hold"] = rng.integers(0, 20000, C_)
    hold["level_max"] = rng.integers(0, 200, C_)
    hold["level_min"] = rng.integers(100, 256, C_)
    probe = np.zeros(C_, capi.CHAN_PROBE)
    probe["run"] = rng.integers(0, 7, C_)
    probe["alarms"] = rng.integers(0, 3, C_)
    return hold, probe


def _run_fused(ctx, form, pk, sizes, codec, radio, C_, F_, stride, hdr, win, d_st, d_info, d_agg, rank):
    if form == capi.PKT_SLOTS:
        ctx.decode_meter_window(form, gu.to_dev(_to_slots(pk, sizes)), None, gu.to_dev(codec), None, C_, F_, 0, 0, d_st, win, info=d_info, agg=d_agg, rank=rank)
    elif form == capi.PKT_PACKED:
        ctx.decode_meter_window(form, gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), None, C_, F_, stride, hdr, d_st, win, info=d_info, agg=d_agg, rank=rank)
    else:
        ctx.decode_meter_window(form, gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), gu.to_dev(radio), C_, F_, stride, 0, d_st, win, info=d_info, agg=d_agg, rank=rank)


def _run_plain(ctx, form, pk, sizes, codec, radio, C_, F_, stride, hdr, d_st, d_info, d_agg, rank):
    if form == capi.PKT_SLOTS:
        ctx.decode_meter_rtp(gu.to_dev(_to_slots(pk, sizes)), gu.to_dev(codec), C_, F_, d_st, info=d_info, agg=d_agg, rank=rank)
    elif form == capi.PKT_PACKED:
        ctx.decode_meter_packets(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), C_, F_, stride, hdr, d_st, info=d_info, agg=d_agg, rank=rank)
    else:
        ctx.decode_meter_packets_mixed(gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), gu.to_dev(radio), C_, F_, stride, d_st, info=d_info, agg=d_agg, rank=rank)


@pytest.mark.parametrize("mode", [capi.GATE_ALWAYS, capi.GATE_SQU, capi.GATE_PTT, capi.GATE_SQU_OR_PTT])
def test_window_update_vs_oracle(ctx, orc, mode):
    """igdsp_window_update on records: per-frame gates from info.ed137, lengths from info.payload_len or from len, EMPTY records,
    a closed per-channel gate, pre-existing hold / probe state, a small alarm length so alarms occur."""
    torch = gu.torch_cuda()
    rng = np.random.default_rng(100 + mode)
    C_, F_, n = 77, 61, 160
    pl = orc.gen_uniform(F_ * C_ * n, seed=mode).reshape(F_, C_, n).copy()
    pl[rng.random((F_, C_)) < 0.6] = 0xD5
    codec = np.where(np.arange(C_) % 4 == 0, 0, 8).astype(np.uint8)
    length = np.where(rng.random((F_, C_)) < 0.15, rng.integers(0, 161, (F_, C_)), 160).astype(np.uint16)
    st = orc.decode_meter(pl, codec, length=length)
    info = np.zeros((F_, C_), capi.RTP_INFO)
    info["ed137"] = rng.integers(0, 1 << 32, (F_, C_), dtype=np.uint64).astype(np.uint32)
    info["ed137"][rng.random((F_, C_)) < 0.3] = 0
    info["payload_len"] = length
    gate = (np.arange(C_) % 5 != 0).astype(np.uint8)
    for use_len, use_info in ((False, True), (True, True), (True, False), (False, False)):
        hold0, probe0 = _start_state(rng, C_)
        d_hold, d_probe = gu.to_dev(hold0), gu.to_dev(probe0)
        win = ctx.window(d_hold, gate_mode=mode, gate=gu.to_dev(gate), probe=d_probe, probe_alarm=4)
        ctx.window_update(gu.to_dev(st), C_, F_, n, win, info=gu.to_dev(info) if use_info else None, length=gu.to_dev(length) if use_len else None)
        torch.cuda.synchronize()
        eh, ep = hold0.copy(), probe0.copy()
        orc.window_update(st, eh, info=info if use_info else None, length=length if use_len else None, n=n, gate_mode=mode, alarm=4, gate=gate, probe=ep)
        assert gu.to_host(d_hold, capi.CHAN_HOLD).tobytes() == eh.tobytes(), (use_len, use_info)
        assert gu.to_host(d_probe, capi.CHAN_PROBE).tobytes() == ep.tobytes(), (use_len, use_info)
        assert ep["alarms"].sum() > probe0["alarms"].sum()
    # mode 0 without gate / probe == igdsp_hold_update
    hold0, _ = _start_state(rng, C_)
    d_a, d_b = gu.to_dev(hold0), gu.to_dev(hold0)
    ctx.window_update(gu.to_dev(st), C_, F_, n, ctx.window(d_a), length=gu.to_dev(length))
    torch.cuda.synchronize()
    eh = hold0.copy()
    orc.window_update(st, eh, length=length, n=n)
    assert gu.to_host(d_a, capi.CHAN_HOLD).tobytes() == eh.tobytes()
    assert ctx.L.igdsp_window_update(ctx.h, d_b.data_ptr(), None, None, C_, F_, n, None, None) == -22


@pytest.mark.parametrize("form,hdr,stride,C_,F_", [
    (capi.PKT_SLOTS, 20, 180, 64, 40), (capi.PKT_SLOTS, 20, 180, 192, 17), (capi.PKT_PACKED, 20, 180, 128, 33), (capi.PKT_PACKED, 12, 172, 64, 64),
    (capi.PKT_PACKED, 20, 200, 64, 9), (capi.PKT_MIXED, 0, 180, 128, 24), (capi.PKT_MIXED, 0, 184, 64, 80), (capi.PKT_PACKED, 20, 180, 96, 2),
    (capi.PKT_SLOTS, 20, 180, 32, 6)])
@pytest.mark.parametrize("blk", ["0", "1"])
def test_fused_window_vs_two_step_route(ctx, orc, monkeypatch, form, hdr, stride, C_, F_, blk):
    """packets -> records + info + aggregate + hold[c] + probe[c] in ONE entry, for every gate mode: the records / info / aggregate
    are byte for byte those of the ungated fused entry, hold and probe equal the oracle's fold of the two-step route.  Both fused
    forms: blk = 0, the windows in the waves' registers (F up to 80 frames splits into several segments per channel group, runs
    chained through the work buffer), and blk = 1, the windows in the block's LDS (up to 80 frames through the 16-slot commit
    ring); C = 96 and 32 (not multiples of 64) take the record-wise fold behind the plain fused kernel."""
    torch = gu.torch_cuda()
    monkeypatch.setenv("IGDSP_WIN_BLK", blk)
    rng = np.random.default_rng(form * 1000 + C_ + F_)
    radio = np.ones(C_, np.uint8) if form == capi.PKT_SLOTS else (np.full(C_, hdr == 20, np.uint8) if form == capi.PKT_PACKED else rng.integers(0, 2, C_).astype(np.uint8))
    codec = np.where(np.arange(C_) % 3 == 0, 0, 8).astype(np.uint8)          # mostly A-law: its digital silence is 0xD5
    pk, sizes = _make_traffic(orc, rng, C_, F_, stride, radio, codec, seed=C_ + F_)
    gate = (np.arange(C_) % 7 != 3).astype(np.uint8)
    d_work = gu.dev_zeros(ctx.window_work_bytes(C_), 0xEE)
    d_st0, d_info0, d_agg0 = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
    _run_plain(ctx, form, pk, sizes, codec, radio, C_, F_, stride, hdr, d_st0, d_info0, d_agg0, 5)
    torch.cuda.synchronize()
    for mode in (capi.GATE_ALWAYS, capi.GATE_SQU, capi.GATE_PTT, capi.GATE_SQU_OR_PTT):
        hold0, probe0 = _start_state(rng, C_)
        d_hold, d_probe = gu.to_dev(hold0), gu.to_dev(probe0)
        d_st, d_info, d_agg = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
        win = ctx.window(d_hold, gate_mode=mode, gate=gu.to_dev(gate), probe=d_probe, work=d_work, probe_alarm=5)
        _run_fused(ctx, form, pk, sizes, codec, radio, C_, F_, stride, hdr, win, d_st, d_info, d_agg, 5)
        torch.cuda.synchronize()
        assert gu.to_host(d_st, np.uint8).tobytes() == gu.to_host(d_st0, np.uint8).tobytes(), mode
        assert gu.to_host(d_info, np.uint8).tobytes() == gu.to_host(d_info0, np.uint8).tobytes(), mode
        assert gu.to_host(d_agg, np.uint8).tobytes() == gu.to_host(d_agg0, np.uint8).tobytes(), mode
        est, einfo, elen, metered, eh, ep = _expected(orc, pk, sizes, radio, codec, mode, 5, gate, hold0, probe0)
        gst = gu.to_host(d_st, capi.FRAME_STATS, (F_, C_))
        gu.assert_stats_equal(gst[metered].reshape(1, -1), est[metered].reshape(1, -1), n=elen[metered].reshape(1, -1))
        assert np.all(gst[~metered]["flags"] == capi.FLAG_EMPTY)
        gh, gp = gu.to_host(d_hold, capi.CHAN_HOLD), gu.to_host(d_probe, capi.CHAN_PROBE)
        for fld in capi.CHAN_HOLD.names:
            assert np.array_equal(gh[fld], eh[fld]), (mode, fld, np.argwhere(gh[fld] != eh[fld])[:4].tolist())
        for fld in capi.CHAN_PROBE.names:
            assert np.array_equal(gp[fld], ep[fld]), (mode, fld, np.argwhere(gp[fld] != ep[fld])[:4].tolist())
        if F_ >= 17:
            assert ep["alarms"].sum() > probe0["alarms"].sum()
            if mode == capi.GATE_ALWAYS or radio.any():                      # SIP legs carry no ED-137 word: SQU / PTT gates stay shut
                assert (eh["count"] > hold0["count"]).any()
    # without a work buffer (and without probe tracking) the entry runs the plain fused kernel + the record-wise fold: same window
    hold0, _ = _start_state(rng, C_)
    d_hold = gu.to_dev(hold0)
    d_st, d_info = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE)
    _run_fused(ctx, form, pk, sizes, codec, radio, C_, F_, stride, hdr, ctx.window(d_hold, gate_mode=capi.GATE_SQU), d_st, d_info, None, 0)
    torch.cuda.synchronize()
    _, _, _, _, eh, _ = _expected(orc, pk, sizes, radio, codec, capi.GATE_SQU, 500, None, hold0, np.zeros(C_, capi.CHAN_PROBE))
    assert gu.to_host(d_hold, capi.CHAN_HOLD).tobytes() == eh.tobytes()


@pytest.mark.parametrize("nseg", ["1", "2", "8"])
def test_fused_window_segment_counts(ctx, orc, monkeypatch, nseg):
    """The launcher's segment rule picks 4 at the headline shape; one segment (several units per wave in rounds: what a launch of
    >= 262 144 channels gets), two and eight (the work buffer's limit) must fold to the same windows and runs."""
    torch = gu.torch_cuda()
    monkeypatch.setenv("IGDSP_WIN_NSEG", nseg)
    monkeypatch.setenv("IGDSP_WIN_BLK", "0")
    rng = np.random.default_rng(int(nseg))
    C_, F_, stride = 3200, 70, 180                     # 50 channel groups
    radio, codec = np.ones(C_, np.uint8), np.where(np.arange(C_) % 2, 8, 0).astype(np.uint8)
    pk, sizes = _make_traffic(orc, rng, C_, F_, stride, radio, codec, seed=int(nseg))
    hold0, probe0 = _start_state(rng, C_)
    d_hold, d_probe = gu.to_dev(hold0), gu.to_dev(probe0)
    d_st, d_info = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE)
    win = ctx.window(d_hold, gate_mode=capi.GATE_SQU_OR_PTT, probe=d_probe, work=gu.dev_zeros(ctx.window_work_bytes(C_), 0xEE), probe_alarm=3)
    ctx.decode_meter_window(capi.PKT_PACKED, gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), None, C_, F_, stride, 20, d_st, win, info=d_info)
    torch.cuda.synchronize()
    est, einfo, elen, metered, eh, ep = _expected(orc, pk, sizes, radio, codec, capi.GATE_SQU_OR_PTT, 3, None, hold0, probe0)
    gu.assert_stats_equal(gu.to_host(d_st, capi.FRAME_STATS, (F_, C_))[metered].reshape(1, -1), est[metered].reshape(1, -1), n=elen[metered].reshape(1, -1))
    assert gu.to_host(d_hold, capi.CHAN_HOLD).tobytes() == eh.tobytes()
    assert gu.to_host(d_probe, capi.CHAN_PROBE).tobytes() == ep.tobytes()


@pytest.mark.parametrize("gpb,C_,F_,alarm", [("4", 1024, 50, 3), ("2", 384, 90, 500), ("1", 192, 300, 7), ("4", 256, 520, 2), ("2", 128, 255, 4), ("1", 64, 1, 1), ("4", 256, 256, 6)])
def test_fused_window_block_form(ctx, orc, monkeypatch, gpb, C_, F_, alarm):
    """The block-owned form at 4 / 2 / 1 channel groups per block (the launcher's own rule only leaves one group per block at
    these sizes), launches longer than the 255 frames its packed LDS counters hold (300 -> 2 parts, 520 -> 3), runs that enter
    the launch close to the alarm length, and the records-free mode."""
    torch = gu.torch_cuda()
    monkeypatch.setenv("IGDSP_WIN_BLK", "1")
    monkeypatch.setenv("IGDSP_WIN_GPB", gpb)
    rng = np.random.default_rng(C_ + F_)
    stride = 180
    radio, codec = np.ones(C_, np.uint8), np.where(np.arange(C_) % 4 == 1, 0, 8).astype(np.uint8)
    pk, sizes = _make_traffic(orc, rng, C_, F_, stride, radio, codec, seed=C_ + F_)
    gate = (np.arange(C_) % 5 != 2).astype(np.uint8)
    hold0, probe0 = _start_state(rng, C_)
    probe0["run"][::3] = max(alarm, 2) - 2                       # two probe frames from the alarm
    est, einfo, elen, metered, eh, ep = _expected(orc, pk, sizes, radio, codec, capi.GATE_SQU_OR_PTT, alarm, gate, hold0, probe0)
    d_st0, d_info0, d_agg0 = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
    _run_plain(ctx, capi.PKT_PACKED, pk, sizes, codec, radio, C_, F_, stride, 20, d_st0, d_info0, d_agg0, 3)
    torch.cuda.synchronize()
    for with_records in (True, False):
        d_hold, d_probe = gu.to_dev(hold0), gu.to_dev(probe0)
        d_st, d_info, d_agg = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
        win = ctx.window(d_hold, gate_mode=capi.GATE_SQU_OR_PTT, gate=gu.to_dev(gate), probe=d_probe, work=gu.dev_zeros(ctx.window_work_bytes(C_), 0xEE), probe_alarm=alarm)
        ctx.decode_meter_window(capi.PKT_PACKED, gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), None, C_, F_, stride, 20,
                                d_st if with_records else None, win, info=d_info if with_records else None, agg=d_agg, rank=3)
        torch.cuda.synchronize()
        if with_records:
            gst = gu.to_host(d_st, capi.FRAME_STATS, (F_, C_))
            gu.assert_stats_equal(gst[metered].reshape(1, -1), est[metered].reshape(1, -1), n=elen[metered].reshape(1, -1))
            assert np.all(gst[~metered]["flags"] == capi.FLAG_EMPTY)
            assert gu.to_host(d_st, np.uint8).tobytes() == gu.to_host(d_st0, np.uint8).tobytes()
            assert gu.to_host(d_info, np.uint8).tobytes() == gu.to_host(d_info0, np.uint8).tobytes()
        assert gu.to_host(d_agg, np.uint8).tobytes() == gu.to_host(d_agg0, np.uint8).tobytes()        # (a launch in parts adds up to the same aggregate)
        gh, gp = gu.to_host(d_hold, capi.CHAN_HOLD), gu.to_host(d_probe, capi.CHAN_PROBE)
        for fld in capi.CHAN_HOLD.names:
            assert np.array_equal(gh[fld], eh[fld]), (with_records, fld, np.argwhere(gh[fld] != eh[fld])[:4].tolist())
        for fld in capi.CHAN_PROBE.names:
            assert np.array_equal(gp[fld], ep[fld]), (with_records, fld, np.argwhere(gp[fld] != ep[fld])[:4].tolist())
    if F_ > 1:
        assert ep["alarms"].sum() > probe0["alarms"].sum()


@pytest.mark.parametrize("blk", ["0", "1"])
def test_fused_window_without_records(ctx, orc, monkeypatch, blk):
    """d_stats == NULL on the fused path: the windows and runs alone, equal to the run that also writes the records."""
    torch = gu.torch_cuda()
    monkeypatch.setenv("IGDSP_WIN_BLK", blk)
    rng = np.random.default_rng(77)
    C_, F_, stride = 128, 40, 180
    radio, codec = np.ones(C_, np.uint8), np.full(C_, 8, np.uint8)
    pk, sizes = _make_traffic(orc, rng, C_, F_, stride, radio, codec, seed=5)
    hold0, probe0 = _start_state(rng, C_)
    outs = []
    for with_records in (True, False):
        d_hold, d_probe = gu.to_dev(hold0), gu.to_dev(probe0)
        win = ctx.window(d_hold, gate_mode=capi.GATE_PTT, probe=d_probe, work=gu.dev_zeros(ctx.window_work_bytes(C_)), probe_alarm=4)
        d_st = gu.dev_zeros(F_ * C_ * 16, 0xEE) if with_records else None
        ctx.decode_meter_window(capi.PKT_PACKED, gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), None, C_, F_, stride, 20, d_st, win)
        torch.cuda.synchronize()
        outs.append((gu.to_host(d_hold, capi.CHAN_HOLD).tobytes(), gu.to_host(d_probe, capi.CHAN_PROBE).tobytes()))
    assert outs[0] == outs[1]
    _, _, _, _, eh, ep = _expected(orc, pk, sizes, radio, codec, capi.GATE_PTT, 4, None, hold0, probe0)
    assert outs[1] == (eh.tobytes(), ep.tobytes())
    # a window without run tracking (d_probe == NULL): the same hold
    d_hold = gu.to_dev(hold0)
    win = ctx.window(d_hold, gate_mode=capi.GATE_PTT, work=gu.dev_zeros(ctx.window_work_bytes(C_)))
    ctx.decode_meter_window(capi.PKT_PACKED, gu.to_dev(pk), gu.to_dev(sizes), gu.to_dev(codec), None, C_, F_, stride, 20, None, win)
    torch.cuda.synchronize()
    assert gu.to_host(d_hold, capi.CHAN_HOLD).tobytes() == eh.tobytes()
    # off the fused path the records are needed
    import ctypes as C
    w = ctx.window(gu.to_dev(hold0))
    p = gu.dev_zeros(1 << 16).data_ptr()
    assert ctx.L.igdsp_decode_meter_window(ctx.h, capi.PKT_PACKED, p, None, p, None, 64, 1, 180, 20, None, None, None, 0, C.byref(w), None) == -22


def test_fused_window_argument_rules(ctx):
    d = gu.dev_zeros(1 << 16)
    p = d.data_ptr()
    L = ctx.L
    import ctypes as C

    w = ctx.window(d)
    assert L.igdsp_decode_meter_window(ctx.h, 3, p, None, p, None, 64, 1, 180, 20, p, None, None, 0, C.byref(w), None) == -22      # layout
    assert L.igdsp_decode_meter_window(ctx.h, capi.PKT_PACKED, p, None, p, None, 64, 1, 180, 16, p, None, None, 0, C.byref(w), None) == -22   # header type
    assert L.igdsp_decode_meter_window(ctx.h, capi.PKT_MIXED, p, None, p, None, 64, 1, 180, 0, p, None, None, 0, C.byref(w), None) == -22    # radio missing
    assert L.igdsp_decode_meter_window(ctx.h, capi.PKT_PACKED, p, None, p, None, 33, 1, 180, 20, p, None, None, 0, C.byref(w), None) == -22   # C*F % 64
    assert L.igdsp_decode_meter_window(ctx.h, capi.PKT_PACKED, p, None, p, None, 64, 1, 180, 20, p, None, None, 0, None, None) == -22         # no window
    bad = capi.Window(7, 0, p, None, None, None)
    assert L.igdsp_decode_meter_window(ctx.h, capi.PKT_PACKED, p, None, p, None, 64, 1, 180, 20, p, None, None, 0, C.byref(bad), None) == -22   # gate mode
    nohold = capi.Window(0, 0, None, None, None, None)
    assert L.igdsp_decode_meter_window(ctx.h, capi.PKT_PACKED, p, None, p, None, 64, 1, 180, 20, p, None, None, 0, C.byref(nohold), None) == -22
    w96 = ctx.window(d, gate_mode=capi.GATE_SQU)
    assert L.igdsp_decode_meter_window(ctx.h, capi.PKT_PACKED, p, None, p, None, 96, 2, 180, 20, p, None, None, 0, C.byref(w96), None) == -22   # C % 64 != 0 needs info
    assert ctx.window_work_bytes(65536) == 8 * 65536 * 48


@pytest.mark.parametrize("blk", ["0", "auto"])
def test_fused_window_full_size_equals_two_step_on_device(ctx, orc, monkeypatch, blk):
    """65 536 ch x 128 packed 180-byte packets (BASELINE configs[2]'s shape behind the depayloader): the fused window launch
    (blk = auto: the block-owned form, 256 blocks of four groups; blk = 0: the register form) against the plain fused launch
    followed by igdsp_window_update on the same device data — records, info, hold and probe byte for byte — plus 2 000 sampled
    channels of hold / probe against the oracle's fold of the device's own records."""
    torch = gu.torch_cuda()
    if blk != "auto":
        monkeypatch.setenv("IGDSP_WIN_BLK", blk)
    C_, F_, stride = 65536, 128, 180
    rng = np.random.default_rng(9)
    d_pk = torch.empty((F_ * C_ * stride,), dtype=torch.uint8, device="cuda")
    ctx.gen_uniform(d_pk, d_pk.numel(), seed=77, stream=torch.cuda.current_stream().cuda_stream)
    v = d_pk.view(F_, C_, stride)
    v[:, :, 0] = 0x90                                                       # V = 2, X
    v[:, :, 1] = 8                                                          # PT 8 (A-law), no marker
    v[:, :, 12] = 0x01; v[:, :, 13] = 0x67; v[:, :, 14] = 0x00; v[:, :, 15] = 0x01   # ED-137 extension, one word
    sil = torch.from_numpy(rng.random((F_, C_)) < 0.7).cuda()               # 70 % digital-silence frames: long probe runs
    v[:, :, 20:][sil] = 0xD5
    codec = torch.full((C_,), 8, dtype=torch.uint8, device="cuda")
    hold0, probe0 = _start_state(rng, C_)
    outs = []
    for fused in (True, False):
        d_hold, d_probe = gu.to_dev(hold0), gu.to_dev(probe0)
        d_st, d_info = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE)
        win = ctx.window(d_hold, gate_mode=capi.GATE_SQU_OR_PTT, probe=d_probe, work=gu.dev_zeros(ctx.window_work_bytes(C_)), probe_alarm=6)
        if fused:
            ctx.decode_meter_window(capi.PKT_PACKED, d_pk, None, codec, None, C_, F_, stride, 20, d_st, win, info=d_info)
        else:
            ctx.decode_meter_packets(d_pk, None, codec, C_, F_, stride, 20, d_st, info=d_info)
            ctx.window_update(d_st, C_, F_, 160, win, info=d_info)
        torch.cuda.synchronize()
        outs.append((d_st, d_info, d_hold, d_probe))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    st = gu.to_host(outs[0][0], capi.FRAME_STATS, (F_, C_))
    info = gu.to_host(outs[0][1], capi.RTP_INFO, (F_, C_))
    pick = np.sort(rng.choice(C_, 2000, replace=False))
    eh, ep = hold0[pick].copy(), probe0[pick].copy()
    orc.window_update(np.ascontiguousarray(st[:, pick]), eh, info=np.ascontiguousarray(info[:, pick]), n=160, gate_mode=capi.GATE_SQU_OR_PTT, alarm=6, probe=ep)
    assert gu.to_host(outs[0][2], capi.CHAN_HOLD)[pick].tobytes() == eh.tobytes()
    assert gu.to_host(outs[0][3], capi.CHAN_PROBE)[pick].tobytes() == ep.tobytes()
    assert ep["alarms"].sum() > probe0[pick]["alarms"].sum() and ((st["flags"] & capi.FLAG_PROBE_D5) != 0).mean() > 0.5


def test_block_forms_over_two_rounds_of_blocks(ctx, orc, monkeypatch):
    """131 072 channels = 512 blocks of four groups: two rounds of blocks on the 256 CUs (the launchers' rule still picks the
    block-owned forms).  Fused window against the two-step route on the device; block-owned round trip against the static form."""
    torch = gu.torch_cuda()
    C_, F_, stride = 131072, 12, 180
    rng = np.random.default_rng(21)
    d_pk = torch.empty((F_ * C_ * stride,), dtype=torch.uint8, device="cuda")
    ctx.gen_uniform(d_pk, d_pk.numel(), seed=5, stream=torch.cuda.current_stream().cuda_stream)
    v = d_pk.view(F_, C_, stride)
    v[:, :, 0] = 0x90; v[:, :, 1] = 8
    v[:, :, 12] = 0x01; v[:, :, 13] = 0x67; v[:, :, 14] = 0x00; v[:, :, 15] = 0x01
    v[:, :, 20:][torch.from_numpy(rng.random((F_, C_)) < 0.6).cuda()] = 0xD5
    codec = torch.full((C_,), 8, dtype=torch.uint8, device="cuda")
    hold0, probe0 = _start_state(rng, C_)
    outs = []
    for fused in (True, False):
        d_hold, d_probe = gu.to_dev(hold0), gu.to_dev(probe0)
        d_st, d_info = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * 8, 0xEE)
        win = ctx.window(d_hold, gate_mode=capi.GATE_SQU_OR_PTT, probe=d_probe, work=gu.dev_zeros(ctx.window_work_bytes(C_)), probe_alarm=3)
        if fused:
            ctx.decode_meter_window(capi.PKT_PACKED, d_pk, None, codec, None, C_, F_, stride, 20, d_st, win, info=d_info)
        else:
            ctx.decode_meter_packets(d_pk, None, codec, C_, F_, stride, 20, d_st, info=d_info)
            ctx.window_update(d_st, C_, F_, 160, win, info=d_info)
        torch.cuda.synchronize()
        outs.append((d_st, d_info, d_hold, d_probe))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert gu.to_host(outs[0][3], capi.CHAN_PROBE)["alarms"].sum() > probe0["alarms"].sum()
    # round trip over the payload part of the same bytes (160-byte frames: reuse the buffer's head)
    n = 160
    pay = d_pk[: F_ * C_ * n]
    rt = []
    for blk in ("1", "0"):
        monkeypatch.setenv("IGDSP_RT_BLK", blk)
        d_out, d_st, d_hold = gu.dev_zeros(F_ * C_ * n, 0xEE), gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.to_dev(hold0)
        ctx.roundtrip_peakhold(pay, codec, C_, F_, n, d_out, d_st, d_hold, variant=capi.ENC_G191)
        torch.cuda.synchronize()
        rt.append((d_out, d_st, d_hold))
    for a, b in zip(rt[0], rt[1]):
        assert torch.equal(a, b)
