#!/usr/bin/env python3
"""Latency of the drop-in per-frame path: stage one 20 ms frame per call with igdsp_on_rtp_frame, then igdsp_flush
(the 40 ms owner-thread tick of the reference).  Prints host-side wall time per flush for 4 / 32 / 64 / 1024 calls."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    for nch in (4, 32, 64, 1024):
        ctx = capi.Context(device=0, max_channels=nch)
        for c in range(nch):
            ctx.map_call(1000 + c, c)
        rng = np.random.default_rng(nch)
        frames = [rng.integers(0, 256, 160, dtype=np.uint8).tobytes() for _ in range(nch)]
        ts, ts_stage = [], []
        for it in range(220):
            t0 = time.perf_counter()
            for c in range(nch):
                ctx.on_rtp_frame(1000 + c, 0, frames[c])
            t1 = time.perf_counter()
            n = ctx.flush()
            t2 = time.perf_counter()
            assert n == nch
            if it >= 20:
                ts.append(t2 - t1)
                ts_stage.append((t1 - t0) / nch)
        lv = ctx.poll(0)
        print(f"{nch:5d} calls: flush median {np.median(ts) * 1e6:7.1f} us  p99 {np.percentile(ts, 99) * 1e6:7.1f} us   "
              f"(staging via ctypes {np.median(ts_stage) * 1e6:.2f} us per frame)  ch0 rms {lv.rms:.1f} peak {lv.peak}")
        ctx.close()


if __name__ == "__main__":
    main()
