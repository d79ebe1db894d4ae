#!/usr/bin/env python3
"""Round-trip kernel on one placed buffer set under experiment knobs (IGDSP_RT_NSEG / IGDSP_RT_ORDER), same process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from igate4xsoftphonedsp_amd import capi
from oracle import oracle as orc
C_, F_, n = 65536, 128, 160
B = C_ * F_ * n
torch.cuda.set_device(0)
ctx = capi.Context(0, 64)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); hs = s.cuda_stream
ioset, (p_in, p_st, p_hold, p_out, p_yard), rep = ctx.io_alloc([(B, capi.IO_INPUT), (F_ * C_ * 16, capi.IO_RECORD), (C_ * 32, capi.IO_RECORD), (B, capi.IO_BULK), (2 * B, capi.IO_BULK)])
print(rep)
d_pl = capi.as_tensor(p_in, B, torch.uint8, (F_, C_, n))
tile = orc.gen_speech(480, F_, n, (np.arange(480) & 1).astype(np.uint8) * 8)
d_pl.copy_(torch.from_numpy(tile).cuda()[:, torch.arange(C_, device="cuda") % 480, :])
cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda"); cd[1::2] = 8
ctx.hold_reset(p_hold, C_, stream=hs)
def ms(reps=40):
    t = ctx.timer(); t.start(hs)
    for _ in range(reps): ctx.roundtrip_peakhold(p_in, cd, C_, F_, n, p_out, p_st, p_hold, stream=hs)
    t.stop(hs); v = t.elapsed_ms() / reps; t.close(); return v
import ctypes as CT
mix = ctx.L.igdsp_internal_stream_mix2
mix.restype = CT.c_int
mix.argtypes = [CT.c_void_p, CT.c_void_p, CT.c_void_p, CT.c_void_p, CT.c_uint32, CT.c_int, CT.c_int, CT.c_int, CT.c_void_p, CT.c_void_p]
def bare(reps=40):
    # the yardstick on the same buffers: 8 KiB read + 8 KiB written per item, odd items into the second half of the output
    n_items = B // 8192
    t = ctx.timer(); t.start(hs)
    for _ in range(reps): mix(ctx.h, p_in, p_yard, p_yard + B, n_items, 8, 8, 12, hs, None)
    t.stop(hs); v = t.elapsed_ms() / reps; t.close(); return v
for _ in range(5): ms()
bare(10)
b = min(bare() for _ in range(3))
print(f"bare 1:1 stream on these buffers: {b:.4f} ms  {2 * B / b / 1e6:.0f} GB/s", flush=True)
ctx.set_variant(4); ms(10); v4 = min(ms() for _ in range(3)); ctx.set_variant(0)
print(f"cell-table form (variant 4): {v4:.4f} ms", flush=True)
for order in (0, 1):
    for nseg in (2, 3, 4, 6):
        os.environ["IGDSP_RT_ORDER"] = str(order); os.environ["IGDSP_RT_NSEG"] = str(nseg)
        ms(10)
        v = min(ms() for _ in range(3))
        print(f"order {order} n_seg {nseg:2d}: {v:.4f} ms  {B * 2.10625 / v / 1e6:.0f} GB/s", flush=True)
os.environ.pop("IGDSP_RT_NSEG")
for rep in range(4):                                   # A/B of the default segment count under both orders, alternating
    for order in (0, 1):
        os.environ["IGDSP_RT_ORDER"] = str(order)
        ms(10)
        v = min(ms() for _ in range(3))
        print(f"A/B {rep}: order {order}: {v:.4f} ms  frac {B * 2.10625 / v / 1e6 / 8000:.4f}", flush=True)
