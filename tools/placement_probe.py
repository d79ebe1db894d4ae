#!/usr/bin/env python3
"""Does the read-stream rate depend on WHICH allocation holds the batch?  Allocates several batch-sized buffers in one
process and times the bare read-stream and read+record kernels on each (separate processes on one box were seen in a
'fast' (0.227 ms) and a 'slow' (0.252 ms) state for the same kernel)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    nbytes = 65536 * 128 * 160
    fn = ctx.L.igdsp_internal_stream_rw
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    sink = torch.zeros((8,), dtype=torch.int64, device="cuda")
    dst = torch.empty((nbytes // 10 + 4096,), dtype=torch.uint8, device="cuda")
    bufs = []
    for i in range(6):
        b = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
        ctx.gen_uniform(b, nbytes, seed=i + 1)
        bufs.append(b)
    big = torch.empty((4 * nbytes,), dtype=torch.uint8, device="cuda")
    ctx.gen_uniform(big, big.numel(), seed=9)
    for k in range(4):
        bufs.append(big[k * nbytes:(k + 1) * nbytes])
    dsts = [dst] + [torch.empty((nbytes // 10 + 4096 + (i << 21),), dtype=torch.uint8, device="cuda") for i in range(1, 4)]
    torch.cuda.synchronize()
    tm = ctx.timer()
    print("dst ptrs", [hex(d.data_ptr()) for d in dsts])
    for i, b in enumerate(bufs):
        row = []
        for d in dsts:
            for _ in range(3):
                fn(ctx.h, b.data_ptr(), nbytes, d.data_ptr(), None)
            tm.start(None)
            for _ in range(20):
                fn(ctx.h, b.data_ptr(), nbytes, d.data_ptr(), None)
            tm.stop(None)
            row.append(tm.elapsed_ms() / 20)
        print(f"buf {i} ptr {b.data_ptr():#x} rw per dst: " + " ".join(f"{x:.4f}" for x in row))


if __name__ == "__main__":
    main()
