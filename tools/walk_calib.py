#!/usr/bin/env python3
"""What does the channel-group-major walk of the fused window kernel cost the memory system by itself?  (GPU box)
A compute-free kernel fetching the packed-packet pieces (k_stream_pieces: ascending items; k_stream_walk: a wave walks the
frames of one channel group, n_seg segments) beside the product kernels, on one igdsp_io_alloc buffer set."""
import ctypes as CT, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
st, ptrs, rep = ctx.io_alloc([(in_b, capi.IO_INPUT), (B * 16, capi.IO_RECORD), (B * 8, capi.IO_RECORD), (C_ * 32, capi.IO_RECORD),
                              (C_ * 8, capi.IO_RECORD), (ctx.window_work_bytes(C_), capi.IO_RECORD)])
print({k: rep[k] for k in ("placed", "classes_found", "chunks_explored", "settle_ms")}, flush=True)
ctx.gen_uniform(ptrs[0], in_b, stream=hs)
pk = capi.as_tensor(ptrs[0], B * 180, torch.uint8, (F_, C_, 180))
pk[:, :, 0] = 0x90; pk[:, :, 1] = 0
ctx.hold_reset(ptrs[3], C_, stream=hs)
ctx.dev_memset(ptrs[4], 0, C_ * 8)
torch.cuda.synchronize()
fp = ctx.L.igdsp_internal_stream_pieces
fp.restype = CT.c_int
fp.argtypes = [CT.c_void_p, CT.c_void_p, CT.c_uint32, CT.c_uint32, CT.c_uint32, CT.c_int, CT.c_int, CT.c_void_p, CT.c_void_p, CT.c_void_p]
fw = ctx.L.igdsp_internal_stream_walk
fw.restype = CT.c_int
fw.argtypes = [CT.c_void_p, CT.c_void_p, CT.c_uint32, CT.c_uint32, CT.c_uint32, CT.c_uint32, CT.c_uint32, CT.c_uint32, CT.c_void_p, CT.c_void_p, CT.c_void_p]
NB = B * (180 + 16 + 8)


def run(label, f, groups=6, per=40):
    for _ in range(20):
        f()
    out = []
    for _ in range(groups):
        t = ctx.timer(); t.start(hs)
        for _ in range(per):
            f()
        t.stop(hs); out.append(t.elapsed_ms() / per); t.close()
    m = statistics.median(out)
    print("%-72s %.4f ms  %.3f of 8 TB/s" % (label, m, NB / (m * 1e-3) / 8e12), flush=True)


for rnd in range(2):
    run("bare pieces, ascending items (k_stream_pieces)", lambda: fp(ctx.h, ptrs[0], B // 64, 180, 20, 0, 12, ptrs[1], ptrs[2], hs))
    for n_seg in (1, 3, 8, 32):
        run("bare walk, channel-group-major, %d segments (k_stream_walk)" % n_seg, lambda n_seg=n_seg: fw(ctx.h, ptrs[0], B // 64, 180, 20, C_ // 64, n_seg, 0, ptrs[1], ptrs[2], hs))
    for tr in (0, 1, 2, 3):
        run("bare, ascending with one item of lookahead, trickle %d" % tr, lambda tr=tr: fw(ctx.h, ptrs[0], B // 64, 180, 20, 0, 1, tr, ptrs[1], ptrs[2], hs))
        run("bare walk, 3 segments,                       trickle %d" % tr, lambda tr=tr: fw(ctx.h, ptrs[0], B // 64, 180, 20, C_ // 64, 3, tr, ptrs[1], ptrs[2], hs))
    run("igdsp_decode_meter_packets", lambda: ctx.decode_meter_packets(ptrs[0], None, cd, C_, F_, 180, 20, ptrs[1], info=ptrs[2], agg=agg, rank=0, stream=hs))
    win = ctx.window(ptrs[3], gate_mode=capi.GATE_SQU_OR_PTT, probe=ptrs[4], work=ptrs[5])
    run("igdsp_decode_meter_window", lambda: ctx.decode_meter_window(capi.PKT_PACKED, ptrs[0], None, cd, None, C_, F_, 180, 20, ptrs[1], win, info=ptrs[2], agg=agg, rank=0, stream=hs))
