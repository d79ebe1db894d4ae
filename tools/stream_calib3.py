#!/usr/bin/env python3
"""Bare read : write mixes on igdsp_io_alloc-placed buffers (inputs class A, bulk output halves in classes B | C), by waves
per CU: what a compute-free kernel moving the same bytes reaches, the yardstick for the write-heavy product kernels."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    fn = ctx.L.igdsp_internal_stream_mix2
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n_items = 131072
    win = n_items * 10 * 1024                       # room for r, w <= 10 pieces per item
    ioset, (p_in, p_out), rep = ctx.io_alloc([(win, capi.IO_INPUT), (2 * win, capi.IO_BULK)])
    print(rep, flush=True)
    ctx.gen_uniform(p_in, win)
    tm = ctx.timer()
    for _ in range(200):
        fn(ctx.h, p_in, p_out, p_out + win, n_items, 8, 8, 16, None, None)
    for r, w in [(8, 8), (8, 4), (4, 8), (10, 1), (8, 2)]:
        for waves in (3, 4, 6, 8, 12, 16):
            res = []
            for d2 in (p_out, p_out + win):           # all writes into the first half (one class) | odd items into the second half (two classes)
                for _ in range(5):
                    assert fn(ctx.h, p_in, p_out, d2, n_items, r, w, waves, None, None) == 0
                tm.start(None)
                for _ in range(30):
                    fn(ctx.h, p_in, p_out, d2, n_items, r, w, waves, None, None)
                tm.stop(None)
                ms = tm.elapsed_ms() / 30
                res.append((ms, n_items * (r + w) * 1024 / ms / 1e6))
            print(f"read{r}:write{w} {waves:2d} waves/CU  one class {res[0][0]:.4f} ms {res[0][1]:.0f} GB/s | two classes {res[1][0]:.4f} ms {res[1][1]:.0f} GB/s", flush=True)
    ioset.close()
    ctx.close()


if __name__ == "__main__":
    main()
