#!/usr/bin/env python3
"""Sweep the record buffer's offset relative to the payload for the bare read+record kernel (see placement_probe.py)."""
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
    src = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    ctx.gen_uniform(src, nbytes, seed=1)
    arena = torch.empty((nbytes // 10 + (96 << 20),), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    tm = ctx.timer()
    print(f"src {src.data_ptr():#x} arena {arena.data_ptr():#x} diff {(arena.data_ptr() - src.data_ptr()) / 2**20:.1f} MiB")
    offs = [0, 1 << 10, 4 << 10, 16 << 10, 64 << 10, 256 << 10, 1 << 20] + [k << 21 for k in range(1, 33)]
    for off in offs:
        d = arena.data_ptr() + off
        for _ in range(3):
            fn(ctx.h, src.data_ptr(), nbytes, d, None)
        tm.start(None)
        for _ in range(20):
            fn(ctx.h, src.data_ptr(), nbytes, d, None)
        tm.stop(None)
        print(f"off {off / 2**20:9.4f} MiB  (dst-src) mod 64MiB = {((d - src.data_ptr()) % (64 << 20)) / 2**20:8.3f}  rw {tm.elapsed_ms() / 20:.4f} ms")


if __name__ == "__main__":
    main()
