"""-m gpu: launches that are NOT synchronised, on two streams of one context, from two host threads, with flushes on the
context's own stream in between.

The persistent kernels draw their batches from a device-side work-counter pair (igdsp_ctx::d_queues).  Round 2 handed
launch k pair k % 64 with no completion check, so with >= 64 launches pending on one stream a launch on another stream
could receive a pair whose owner had not run yet: two kernels on one counter, super-chunks skipped in one and out-of-range
ids in the other — wrong records with rc 0.  Pairs are now keyed by stream (launches of one stream serialise); this test
keeps > 64 launches pending on each stream behind a block of long launches and checks every record, PCM byte, re-encoded
byte, hold window and aggregate of every job against the oracle."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from igate4xsoftphonedsp_amd import capi  # noqa: E402
from tests import gpu_util as gu  # noqa: E402

N_JOBS = 220            # per the round-2 review: >= 200 un-synchronised launches of mixed sizes
N_BLOCKERS = 40         # long launches at the head of each stream so that everything behind them is pending at once


def _make_jobs(orc, rng):
    """job = dict(kind, device inputs, device outputs, expected numpy results); inputs are uploaded (synchronously) up front."""
    jobs = []
    for it in range(N_JOBS):
        kind = ("meter", "pcm", "strided", "roundtrip", "strided_pcm", "meter")[it % 6]
        if kind in ("meter", "pcm"):
            C_, F_, n = int(rng.choice([64, 200, 1024, 4096, 16384])), int(rng.integers(1, 9)), 160
        elif kind in ("strided", "strided_pcm"):
            C_, F_, n = int(rng.choice([64, 640, 2048])), int(rng.integers(1, 7)), int(rng.choice([164, 24, 240, 80]))
        else:
            C_, F_, n = int(rng.choice([64, 128, 1024, 4160])), int(rng.choice([2, 8, 24])), 160
        pl = orc.gen_uniform(F_ * C_ * n, seed=5000 + it).reshape(F_, C_, n)
        cd = rng.choice(np.array([0, 8], np.uint8), size=C_)
        j = {"kind": kind, "C": C_, "F": F_, "n": n, "pl": gu.to_dev(pl), "cd": gu.to_dev(cd), "rank": it % 8,
             "st": gu.dev_zeros(F_ * C_ * 16, 0xEE)}
        if kind in ("meter", "strided"):
            j["agg"] = gu.dev_zeros(capi.AGGREGATE.itemsize)
            j["e_st"], j["e_agg"] = orc.decode_meter(pl, cd, want_agg=True, rank=j["rank"])
        elif kind in ("pcm", "strided_pcm"):
            j["pcm"] = gu.dev_zeros(F_ * C_ * n * 2, 0xEE)
            j["e_st"], j["e_pcm"] = orc.decode_meter(pl, cd, want_pcm=True)
        else:
            j["out"] = gu.dev_zeros(F_ * C_ * n, 0xEE)
            j["hold"] = gu.to_dev(gu.new_hold(C_))
            j["variant"] = it & 1
            j["e_out"], j["e_st"], j["e_hold"] = orc.roundtrip_peakhold(pl, cd, orc.hold_new(C_), variant=j["variant"])
        jobs.append(j)
    return jobs


def _enqueue(ctx, j, hs):
    if j["kind"] in ("meter", "strided"):
        ctx.decode_meter(j["pl"], j["cd"], j["C"], j["F"], j["n"], j["st"], agg=j["agg"], rank=j["rank"], stream=hs)
    elif j["kind"] in ("pcm", "strided_pcm"):
        ctx.decode_meter(j["pl"], j["cd"], j["C"], j["F"], j["n"], j["st"], pcm=j["pcm"], stream=hs)
    else:
        ctx.roundtrip_peakhold(j["pl"], j["cd"], j["C"], j["F"], j["n"], j["out"], j["st"], j["hold"], variant=j["variant"], stream=hs)


def _check(j):
    C_, F_, n = j["C"], j["F"], j["n"]
    gu.assert_stats_equal(gu.to_host(j["st"], capi.FRAME_STATS, (F_, C_)), j["e_st"], n=n)
    if "agg" in j:
        assert gu.to_host(j["agg"], capi.AGGREGATE)[0].tobytes() == j["e_agg"].tobytes(), (j["kind"], C_, F_, n)
    if "pcm" in j:
        assert np.array_equal(gu.to_host(j["pcm"], "<i2", (F_, C_, n)), j["e_pcm"]), (j["kind"], C_, F_, n)
    if "out" in j:
        assert np.array_equal(gu.to_host(j["out"], np.uint8, (F_, C_, n)), j["e_out"])
        assert gu.to_host(j["hold"], capi.CHAN_HOLD).tobytes() == j["e_hold"].tobytes()


def test_unsynchronised_launches_on_two_streams_and_two_threads(orc):
    torch = gu.torch_cuda()
    ctx = capi.Context(0, 256)
    rng = np.random.default_rng(20241218)
    jobs = _make_jobs(orc, rng)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    # the long launches at the head of each stream: 16 384 ch x 128 frames (0.34 GB, ~60 us each), verified against one
    # synchronous run of the same launch
    Cb, Fb = 16384, 128
    d_big = torch.empty((Fb * Cb * 160,), dtype=torch.uint8, device="cuda")
    ctx.gen_uniform(d_big, d_big.numel(), stream=torch.cuda.current_stream().cuda_stream)
    d_bcd = torch.zeros((Cb,), dtype=torch.uint8, device="cuda")
    st_ref = gu.dev_zeros(Fb * Cb * 16, 0xEE)
    ctx.decode_meter(d_big, d_bcd, Cb, Fb, 160, st_ref, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st_blk = [gu.dev_zeros(Fb * Cb * 16, 0xEE) for _ in streams]
    torch.cuda.synchronize()                            # every fill ran on torch's stream: done before the other streams write

    # the drop-in path rides along on the context's own stream: 200 calls, frames staged and flushed while the two threads enqueue
    nch = 200
    for ch in range(nch):
        ctx.map_call(300 + ch, ch)
    hold = orc.hold_new(nch)
    errors = []

    def worker(k):
        try:
            hs = streams[k].cuda_stream
            for _ in range(N_BLOCKERS):
                ctx.decode_meter(d_big, d_bcd, Cb, Fb, 160, st_blk[k], stream=hs)
            for j in jobs[k::2]:                       # the jobs alternate between the two streams
                _enqueue(ctx, j, hs)
        except Exception as e:                          # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for tick in range(6):                               # flushes on ctx->stream between / during the enqueues
        for ch in range(nch):
            pl = orc.gen_uniform(160, seed=90000 + 1000 * tick + ch)
            assert ctx.on_rtp_frame(300 + ch, 0, pl.tobytes()) == 0
            orc.hold_update(orc.decode_meter(pl.reshape(1, 1, -1), [0]), 160, hold[ch:ch + 1])
        assert ctx.flush() == nch
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()

    for k in range(2):
        assert torch.equal(st_blk[k], st_ref), f"long launches on stream {k} differ from the synchronous run"
    for j in jobs:
        _check(j)
    for ch in range(nch):
        h = ctx.get_hold(ch)
        for f in capi.CHAN_HOLD.names:
            assert int(h[f]) == int(hold[f][ch]), (f, ch)
    assert ctx.L.igdsp_last_error(ctx.h) in (b"", None)
    ctx.close()


def test_many_streams_share_no_work_counters(orc):
    """More streams than the context has work-counter pairs (64): the launches past the table take the static schedule, and a
    stream that igdsp_sync has seen idle gives its pair back.  Every launch still equals the oracle."""
    torch = gu.torch_cuda()
    ctx = capi.Context(0, 64)
    C_, F_ = 1024, 4
    pl = orc.gen_uniform(F_ * C_ * 160, seed=42).reshape(F_, C_, 160)
    cd = np.where(np.arange(C_) & 1, 8, 0).astype(np.uint8)
    est, eagg = orc.decode_meter(pl, cd, want_agg=True)
    d_pl, d_cd = gu.to_dev(pl), gu.to_dev(cd)
    streams = [torch.cuda.Stream() for _ in range(80)]
    outs = [(gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)) for _ in range(100)]
    torch.cuda.synchronize()                            # the fills ran on torch's stream: done before any other stream adds into them
    for s, (d_st, d_agg) in zip(streams, outs):
        for _ in range(3):                              # the aggregate ADDS: three launches, three times the sums
            ctx.decode_meter(d_pl, d_cd, C_, F_, 160, d_st, agg=d_agg, stream=s.cuda_stream)
    for s in streams[:40]:
        ctx.sync(s.cuda_stream)                         # idle: their pairs return to the table
    for s, (d_st, d_agg) in zip(streams[60:], outs[80:]):
        for _ in range(3):
            ctx.decode_meter(d_pl, d_cd, C_, F_, 160, d_st, agg=d_agg, stream=s.cuda_stream)
    torch.cuda.synchronize()
    for d_st, d_agg in outs:
        gu.assert_stats_equal(gu.to_host(d_st, capi.FRAME_STATS, (F_, C_)), est, n=160)
        a = gu.to_host(d_agg, capi.AGGREGATE)[0]
        for f in ("sumsq", "samples", "frames", "n_silent", "n_clipped", "byte_mean_sum"):
            assert int(a[f]) == 3 * int(eagg[f]), f
        assert int(a["peak_slot"][0]) == int(eagg["peak_slot"][0])
    ctx.close()


def test_flush_pool_with_concurrent_producers(orc):
    """32 768 calls (the snapshot is shared out to the helper-thread pool from 16 384 channels on), four producer threads staging
    frames for disjoint call ranges WHILE the owner thread runs flush_begin / flush_end in a loop: no frame may be lost, counted twice
    or folded into the wrong call — every call's window count, level sum and sum of squares must equal what its own frames give
    (every call stages the same known payload sequence, so the expected window is the oracle's over that sequence)."""
    import ctypes as CT

    nch, per_thread, rounds, fpc = 32768, 8192, 12, 2
    c = capi.Context(device=0, max_channels=nch)
    try:
        for ch in range(nch):
            c.map_call(ch, ch)
        rng = np.random.default_rng(11)
        n_pay = 64
        pool = rng.integers(0, 256, (n_pay, 160), dtype=np.uint8)
        buf = pool.ctypes.data_as(CT.c_void_p)
        fn = c.L.igdsp_internal_stage_many
        fn.restype = CT.c_int
        fn.argtypes = [CT.c_void_p, CT.c_int32, CT.c_uint32, CT.c_uint32, CT.c_uint8, CT.c_void_p, CT.c_uint32, CT.c_uint32]
        est = orc.decode_meter(pool.reshape(n_pay, 1, 160), [0])[:, 0]            # record of each payload of the pool
        bad = []

        def producer(t):
            for r in range(rounds):
                # frame f of call k in this round = pool[(f * n_calls + k) % n_pay] (igdsp_internal_stage_many's rule)
                if fn(c.h, t * per_thread, per_thread, fpc, 0, buf, n_pay, 160) != 0:
                    bad.append((t, r))
                    return
                time.sleep(0.002)

        ths = [threading.Thread(target=producer, args=(t,)) for t in range(4)]
        for t in ths:
            t.start()
        flushed = 0
        while any(t.is_alive() for t in ths):
            flushed += c.flush_begin()
            c.poll(123)                                                       # reads the published set while the flush is open
            assert c.flush_end(wait=True) == 0
        for t in ths:
            t.join()
        flushed += c.flush()
        assert not bad, f"staging overflowed (the owner fell more than 8 frames behind): {bad[:3]}"
        assert flushed == nch * rounds * fpc
        for ch in list(range(0, nch, 997)) + [nch - 1]:
            k = ch % per_thread
            idx = [(f * per_thread + k) % n_pay for f in range(fpc)] * rounds
            h = c.get_hold(ch)
            assert int(h["count"]) == rounds * fpc, ch
            assert int(h["level_sum"]) == int(sum(int(est["byte_mean"][i]) for i in idx)), ch
            assert int(h["sumsq_acc"]) == int(sum(int(est["sumsq"][i]) for i in idx)), ch
            assert int(h["peak_hold"]) == max(int(est["peak"][i]) for i in idx), ch
            assert c.poll(ch).frames == rounds * fpc and c.poll(ch).dropped == 0
    finally:
        c.close()
