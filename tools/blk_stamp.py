#!/usr/bin/env python3
"""Block-owned window kernel (k_meter_rtp64<WIN = 2>): when does each block finish?  Needs a -DIGDSP_BLK_STAMP build (tools/ab.sh style:
IGDSP_CXXFLAGS=-DIGDSP_BLK_STAMP python -m igate4xsoftphonedsp_amd.build --force): every block leaves {start, end of its items,
end of its fold, XCC id} (100 MHz clock) behind the masks in the work buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from igate4xsoftphonedsp_amd import capi

C_, F_ = 65536, 128
B = C_ * F_
torch.cuda.set_device(0)
ctx = capi.Context(0, 1024)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); hs = s.cuda_stream
cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
agg = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")
in_b = B * 180 + (1 << 20)
wb = ctx.window_work_bytes(C_)
st, ptrs, rep = ctx.io_alloc([(in_b, capi.IO_INPUT), (B * 16, capi.IO_RECORD), (B * 8, capi.IO_RECORD), (C_ * 32, capi.IO_RECORD),
                              (C_ * 8, capi.IO_RECORD), (wb, capi.IO_RECORD)])
ctx.gen_uniform(ptrs[0], in_b, stream=hs)
pk = capi.as_tensor(ptrs[0], B * 180, torch.uint8, (F_, C_, 180))
pk[:, :, 0] = 0x90; pk[:, :, 1] = 0
ctx.hold_reset(ptrs[3], C_, stream=hs)
ctx.dev_memset(ptrs[4], 0, C_ * 8)
torch.cuda.synchronize()
win = ctx.window(ptrs[3], gate_mode=capi.GATE_SQU_OR_PTT, probe=ptrs[4], work=ptrs[5])
work = capi.as_tensor(ptrs[5], wb, torch.int32, (wb // 16, 4))
G = C_ // 64 // 4
launch = lambda: ctx.decode_meter_window(capi.PKT_PACKED, ptrs[0], None, cd, None, C_, F_, 180, 20, ptrs[1], win, info=ptrs[2], agg=agg, rank=0, stream=hs)
for _ in range(20):
    launch()
for rep in range(4):
    tm = ctx.timer(); tm.start(hs)
    for _ in range(40):
        launch()
    tm.stop(hs); print("event time per launch: %.4f ms" % (tm.elapsed_ms() / 40)); tm.close()
for it in range(30, 40):
    for _ in range(30):                 # back to back: the stamps of the last launch of a train, at the clocks of sustained load
        launch()
    torch.cuda.synchronize()
    t = work[C_ // 64 * F_: C_ // 64 * F_ + G].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    t0 = t[:, 0].min()
    a, b, c, x = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0, t[:, 3] & 15
    print("start us: max %.1f | items done us: min %.1f mean %.1f p90 %.1f max %.1f | fold us: mean %.2f max %.2f | all done max %.1f" %
          (a.max(), b.min(), b.mean(), np.percentile(b, 90), b.max(), (c - b).mean(), (c - b).max(), c.max()))
    if it == 39:
        for xc in range(8):
            m = x == xc
            if m.any():
                print("  xcc %d: %3d blocks, items done mean %.1f max %.1f" % (xc, m.sum(), b[m].mean(), b[m].max()))
