#!/usr/bin/env python3
"""In-process A/B of headline-kernel variants with controlled placement (payload and records 72 GiB apart inside one
arena = different memory regions): default (device-wide work queue), static per-block queue, fat waves, no aggregate, and the bare
same-traffic stream.  Interleaved groups, HIP-event timed, warm clocks."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)          # default: device-wide work queue
    os.environ["IGDSP_GLOBAL_QUEUE"] = "0"
    ctx_q = capi.Context(device=0, max_channels=64)        # static per-block batches
    del os.environ["IGDSP_GLOBAL_QUEUE"]
    C_, F_, n = 65536, 128, 160
    nbytes = C_ * F_ * n
    arena = torch.empty((nbytes + (124 << 30),), dtype=torch.uint8, device="cuda")
    pl = arena[:nbytes]
    ctx.gen_uniform(pl, nbytes, seed=0x20241218)
    cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
    agg = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")
    fn = ctx.L.igdsp_internal_stream_rw
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    tm = ctx.timer()

    def timed(f, reps=25):
        for _ in range(3):
            f()
        tm.start(None)
        for _ in range(reps):
            f()
        tm.stop(None)
        return tm.elapsed_ms() / reps

    # pick the record position (every 12 GiB) with the fastest bare stream
    best = None
    for k in range(11):
        off = nbytes + (k * 12 << 30)
        off = (off + 4095) & ~4095
        st = arena[off:off + F_ * C_ * 16]
        t = timed(lambda: fn(ctx.h, pl.data_ptr(), nbytes, st.data_ptr(), None), 10)
        print(f"records at +{k * 12} GiB: bare stream {t:.4f} ms")
        if best is None or t < best[0]:
            best = (t, st)
    st = best[1].view(torch.int64)
    variants = {
        "default": lambda: ctx.decode_meter(pl, cd, C_, F_, n, st, agg=agg),
        "static-queue": lambda: ctx_q.decode_meter(pl, cd, C_, F_, n, st, agg=agg),
        "no-agg": lambda: ctx.decode_meter(pl, cd, C_, F_, n, st, agg=None),
        "bare-stream": lambda: fn(ctx.h, pl.data_ptr(), nbytes, st.data_ptr(), None),
    }
    res = {k: [] for k in variants}
    for _ in range(8):
        for k, f in variants.items():
            res[k].append(timed(f))
    for k, v in res.items():
        print(f"{k:14s} median {float(np.median(v)):.4f} ms  " + " ".join("%.4f" % x for x in v))
    ctx.set_variant(3)
    print("fat waves (variant 3): %.4f ms" % timed(lambda: ctx.decode_meter(pl, cd, C_, F_, n, st, agg=agg)))
    ctx.set_variant(0)


if __name__ == "__main__":
    main()
