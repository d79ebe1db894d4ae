#!/usr/bin/env python3
"""Read+record stream time as a function of where BOTH the payload window and the record window sit inside one
large allocation (8 GiB grid)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    win = 65536 * 128 * 160
    total = 80 << 30
    big = torch.empty((total,), dtype=torch.uint8, device="cuda")
    fn = ctx.L.igdsp_internal_stream_rw
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    tm = ctx.timer()
    grid = [g << 30 for g in range(0, 80, 8)]
    print("rows: payload window offset; columns: record window offset (GiB): " + " ".join(f"{g >> 30:5d}" for g in grid))
    for a in grid:
        row = []
        for b in grid:
            bb = b + (4 << 30)                       # record window in the middle of its 8 GiB cell, never overlapping the payload window
            for _ in range(2):
                fn(ctx.h, big.data_ptr() + a, win, big.data_ptr() + bb, None)
            tm.start(None)
            for _ in range(6):
                fn(ctx.h, big.data_ptr() + a, win, big.data_ptr() + bb, None)
            tm.stop(None)
            row.append(tm.elapsed_ms() / 6)
        print(f"{a >> 30:3d} GiB: " + " ".join("%.3f" % x for x in row), flush=True)


if __name__ == "__main__":
    main()
