#!/usr/bin/env python3
"""Cost of dword-aligned (not 16-byte aligned) dwordx4 loads: the 10:1 bare stream with the source shifted by 0/4/8/12 bytes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    fn = ctx.L.igdsp_internal_stream_mix
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    n_items = 131072
    nb = n_items * 10 * 1024
    arena = torch.empty((nb + (124 << 30),), dtype=torch.uint8, device="cuda")
    src = arena[:nb + 4096]
    best = min(((ctx.probe_placement(src, nb, out=arena[nb + (k * 12 << 30) + 8192:][:nb // 10 + 4096], reps=4), k) for k in range(1, 11)))
    dst = arena[nb + (best[1] * 12 << 30) + 8192:]
    tm = ctx.timer()
    for sh in (0, 4, 8, 12, 0):
        for _ in range(3):
            assert fn(ctx.h, src.data_ptr() + sh, dst.data_ptr(), n_items, 10, 1, 16, None) == 0
        tm.start(None)
        for _ in range(20):
            fn(ctx.h, src.data_ptr() + sh, dst.data_ptr(), n_items, 10, 1, 16, None)
        tm.stop(None)
        ms = tm.elapsed_ms() / 20
        print(f"source shifted by {sh:2d} bytes: {ms:.4f} ms  {n_items * 11 * 1024 / ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
