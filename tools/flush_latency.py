#!/usr/bin/env python3
"""Latency of the drop-in per-frame path at the reference's cadence: every call stages TWO 20 ms frames with
igdsp_on_rtp_frame between two ticks, then the owner thread's igdsp_flush runs (the 40 ms timer, roip_ed137.cpp:1756).
Prints host-side wall time per flush and per staged frame for 4 / 32 / 1 024 / 65 536 calls (staging in one native loop,
igdsp_internal_stage_many, so Python's call overhead is not in the figure), and — what an owner thread that must not wait pays —
the time inside igdsp_flush_begin (snapshot + enqueue) and inside the igdsp_flush_end that follows a tick later."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    fpc = 2
    for nch in (4, 32, 1024, 65536):
        ctx = capi.Context(device=0, max_channels=nch)
        for c in range(nch):
            ctx.map_call(c, c)
        rng = np.random.default_rng(nch)
        pool = rng.integers(0, 256, (4096, 160), dtype=np.uint8)
        buf = pool.ctypes.data_as(C.c_void_p)
        fn = ctx.L.igdsp_internal_stage_many
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint8, C.c_void_p, C.c_uint32, C.c_uint32]
        ts, ts_stage, tb, te, tdev = [], [], [], [], []
        iters = 220 if nch <= 1024 else 40
        for it in range(iters):
            t0 = time.perf_counter()
            assert fn(ctx.h, 0, nch, fpc, 0, buf, 4096, 160) == 0
            t1 = time.perf_counter()
            n = ctx.flush()
            t2 = time.perf_counter()
            assert n == nch * fpc
            if it >= 10:
                ts.append(t2 - t1)
                ts_stage.append((t1 - t0) / (nch * fpc))
        # the non-blocking pair: begin now, end after the device has had its time (a real owner thread comes back 40 ms later)
        for it in range(iters):
            assert fn(ctx.h, 0, nch, fpc, 0, buf, 4096, 160) == 0
            t1 = time.perf_counter()
            n = ctx.flush_begin()
            t2 = time.perf_counter()
            assert n == nch * fpc
            while ctx.flush_end(wait=False) != 0:       # how long the device side takes (polled, not part of the owner's time)
                pass
            t3 = time.perf_counter()
            assert fn(ctx.h, 0, nch, fpc, 0, buf, 4096, 160) == 0
            n = ctx.flush_begin()
            time.sleep(0.005)
            t4 = time.perf_counter()
            assert ctx.flush_end(wait=True) == 0
            t5 = time.perf_counter()
            if it >= 10:
                tb.append(t2 - t1); tdev.append(t3 - t2); te.append(t5 - t4)
        lv, h = ctx.poll(0), ctx.get_hold(0)
        assert lv.frames == 3 * iters * fpc == int(h["count"]) and lv.dropped == 0
        print(f"{nch:6d} calls x {fpc} frames: flush median {np.median(ts) * 1e6:9.1f} us  p99 {np.percentile(ts, 99) * 1e6:9.1f} us   "
              f"staging {np.median(ts_stage) * 1e9:6.0f} ns per frame   ch0 rms {lv.rms:.1f} peak {lv.peak} frames {lv.frames}\n"
              f"        owner thread, non-blocking: flush_begin median {np.median(tb) * 1e6:8.1f} us  p99 {np.percentile(tb, 99) * 1e6:8.1f} us; "
              f"device done {np.median(tdev) * 1e6:8.1f} us later; flush_end (5 ms later) median {np.median(te) * 1e6:6.1f} us", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
