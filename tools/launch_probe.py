#!/usr/bin/env python3
"""Diagnostic: fixed launch cost of each kernel on tiny inputs (HIP events, 50 launches each)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from igate4xsoftphonedsp_amd import capi

ctx = capi.Context(0, 1024)
s = torch.cuda.current_stream().cuda_stream
C_, n = 65536, 160
FM = 128
d_pl = torch.empty((FM * C_ * n,), dtype=torch.uint8, device="cuda")
ctx.gen_uniform(d_pl, d_pl.numel(), stream=s)
d_cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
d_st = torch.zeros((FM * C_ * 2,), dtype=torch.int64, device="cuda")
d_agg = torch.zeros((14,), dtype=torch.int64, device="cuda")
sink = torch.zeros((1,), dtype=torch.int64, device="cuda")
tm = ctx.timer()


def timeit(name, fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    tm.start(s)
    for _ in range(reps):
        fn()
    tm.stop(s)
    print(f"{name:55s} {tm.elapsed_ms() / reps * 1e3:9.1f} us/launch")


timeit("stream_read 1 MiB (256x1024 thr, no LDS)", lambda: ctx.stream_read(d_pl, 1 << 20, sink, stream=s))
timeit("stream_read 84 MB", lambda: ctx.stream_read(d_pl, 8 * C_ * n, sink, stream=s))
for F_ in (1, 8, 32, 128):
    for v in (2,):
        ctx.set_variant(v)
        timeit(f"decode_meter variant {v} F={F_} ({F_ * C_ * n / 1e6:.0f} MB)", lambda: ctx.decode_meter(d_pl, d_cd, C_, F_, n, d_st, stream=s))
ctx.set_variant(2)
timeit("decode_meter variant 2 F=128 WITH agg", lambda: ctx.decode_meter(d_pl, d_cd, C_, 128, n, d_st, agg=d_agg, stream=s), reps=20)
timeit("decode_meter variant 2 C=4096 F=1", lambda: ctx.decode_meter(d_pl, d_cd, 4096, 1, n, d_st, stream=s))
timeit("decode_meter variant 2 C=32 F=1 (1 wave)", lambda: ctx.decode_meter(d_pl, d_cd, 32, 1, n, d_st, stream=s))
