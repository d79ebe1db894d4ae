#!/usr/bin/env python3
"""Diagnostic: where do the chunk32 waves spend their cycles?  (uses the non-ABI
igdsp_internal_diag_chunk32 entry; never part of a timed or shipped path)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from igate4xsoftphonedsp_amd import capi

C_, F_, n = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 160
ctx = capi.Context(0, 1024)
L = ctx.L
L.igdsp_internal_diag_chunk32.restype = C.c_int
L.igdsp_internal_diag_chunk32.argtypes = [C.c_void_p] * 3 + [C.c_uint32] * 2 + [C.c_void_p] * 3
d_pl = torch.empty((F_ * C_ * n,), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ctx.gen_uniform(d_pl, d_pl.numel(), stream=s)
d_cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
d_st = torch.zeros((F_ * C_ * 2,), dtype=torch.int64, device="cuda")
nw = 256 * 16
d_dg = torch.zeros((nw * 12,), dtype=torch.int64, device="cuda")
for _ in range(3):
    rc = L.igdsp_internal_diag_chunk32(ctx.h, d_pl.data_ptr(), d_cd.data_ptr(), C_, F_, d_st.data_ptr(), d_dg.data_ptr(), s)
    assert rc == 0
torch.cuda.synchronize()
d = d_dg.cpu().numpy().view(np.uint64).reshape(nw, 12).astype(np.float64)
rt = (d[:, 9] - d[:, 8])
print(f"realtime (100 MHz) per wave: mean {rt.mean() / 100:.1f} us; kernel span by realtime {(d[:, 9].max() - d[:, 8].min()) / 100:.1f} us; begin spread {(d[:, 8].max() - d[:, 8].min()) / 100:.1f} us; end spread {(d[:, 9].max() - d[:, 9].min()) / 100:.1f} us")
print(f"shader clock while alive: {np.mean((d[:, 2] - d[:, 0]) / rt) * 100:.0f} MHz")
t0 = d[:, 0].min()
life = d[:, 2] - d[:, 0]
print(f"kernel span (first begin -> last end): {d[:, 2].max() - t0:.0f} cyc")
print(f"wave begin spread: {d[:, 0].max() - t0:.0f}   lut fill+barrier: mean {np.mean(d[:, 1] - d[:, 0]):.0f}")
print(f"wave lifetime: mean {life.mean():.0f} min {life.min():.0f} max {life.max():.0f}")
it = d[:, 5].mean()
print(f"iterations/wave {it:.0f}; per iteration: setup(grab,law) {np.mean(d[:, 3]) / it:.0f}  half X {np.mean(d[:, 4]) / it:.0f}  half Y {np.mean(d[:, 6]) / it:.0f}  frame-reduce {np.mean(d[:, 10]) / it:.0f}  total {life.mean() / it:.0f}")
for x in range(8):
    m = d[:, 7] == x
    if m.any():
        print(f"  xcc {x}: waves {m.sum()} mean life {life[m].mean():.0f} mean wait/iter {np.mean(d[m, 3]) / it:.0f}")
