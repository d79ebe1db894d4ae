#!/usr/bin/env python3
"""Are the packed-packet / other-frame-size kernels bound by their bytes or by their instructions?  (GPU box)
Times a compute-free kernel that fetches exactly the dword-aligned 16-byte pieces those kernels fetch (k_stream_pieces,
igdsp_internal_stream_pieces) beside the product kernels, on one igdsp_io_alloc buffer set."""
import ctypes as CT, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igate4xsoftphonedsp_amd import capi

C_, F_ = 65536, 128
B = C_ * F_
NMAX = 256                                             # every shape below reads at most B * NMAX bytes of the input buffer
torch.cuda.set_device(0)
ctx = capi.Context(0, 1024)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); hs = s.cuda_stream
cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
agg = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")
in_b = B * NMAX + (1 << 20)
st, ptrs, rep = ctx.io_alloc([(in_b, capi.IO_INPUT), (B * 16, capi.IO_RECORD), (B * 8, capi.IO_RECORD)])
print({k: rep[k] for k in ("placed", "classes_found", "chunks_explored", "settle_ms")}, flush=True)
ctx.gen_uniform(ptrs[0], in_b, stream=hs)
fn = ctx.L.igdsp_internal_stream_pieces
fn.restype = CT.c_int
fn.argtypes = [CT.c_void_p, CT.c_void_p, CT.c_uint32, CT.c_uint32, CT.c_uint32, CT.c_int, CT.c_int, CT.c_void_p, CT.c_void_p, CT.c_void_p]


def run(label, f, nbytes, groups=8, per=50):
    for _ in range(30):
        f()
    out = []
    for _ in range(groups):
        t = ctx.timer(); t.start(hs)
        for _ in range(per):
            f()
        t.stop(hs); out.append(t.elapsed_ms() / per); t.close()
    m = statistics.median(out)
    print("%-58s %.4f ms  %.3f of 8 TB/s" % (label, m, nbytes / (m * 1e-3) / 8e12), flush=True)


def bare(mode, rows, stride, hdr, info):
    assert stride <= NMAX
    def f():
        assert fn(ctx.h, ptrs[0], B // 64, stride, hdr, mode, rows, ptrs[1], ptrs[2] if info else None, hs) == 0
    return f


run("bare pieces: packed 180-byte packets (12 pieces, + info)", bare(0, 12, 180, 20, True), B * (180 + 16 + 8))
run("bare pieces: 164-byte frames (11 pieces)", bare(1, 11, 164, 0, False), B * (164 + 16))
run("bare pieces: 168-byte frames (11 pieces)", bare(1, 11, 168, 0, False), B * (168 + 16))
run("bare pieces: 160-byte frames (10 pieces)", bare(1, 10, 160, 0, False), B * (160 + 16))
for n in (164, 168, 240, 192, 128, 96, 80, 64, 24, 20, 160):
    run("igdsp_decode_meter n = %d" % n, lambda n=n: ctx.decode_meter(ptrs[0], cd, C_, F_, n, ptrs[1], agg=agg, rank=0, stream=hs), B * (n + 17))
pk = capi.as_tensor(ptrs[0], B * 180, torch.uint8, (F_, C_, 180))
pk[:, :, 0] = 0x90; pk[:, :, 1] = 0
torch.cuda.synchronize()
run("igdsp_decode_meter_packets, 180-byte packets", lambda: ctx.decode_meter_packets(ptrs[0], None, cd, C_, F_, 180, 20, ptrs[1], info=ptrs[2], agg=agg, rank=0, stream=hs), B * (180 + 1 + 16 + 8))
