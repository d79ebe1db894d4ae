#!/usr/bin/env python3
"""Write-heavy mixes with the writes in ONE other memory class vs spread over TWO other classes."""
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
    nb = 131072 * 10 * 1024
    arena = torch.empty((240 << 30,), dtype=torch.uint8, device="cuda")
    src = arena[:nb]
    # classify 8 GiB cells against the source window and against each other with the 10:1 probe
    cells = [(g << 30) for g in range(8, 232, 8)]
    t_src = {o: ctx.probe_placement(src, nb, out=arena[o:o + nb // 10 + 4096], reps=4) for o in cells}
    fast = [o for o in cells if t_src[o] < 0.235]
    a = fast[0]
    # a cell of another class than `a` (and than src): probing reads from cell a, writes to cell o
    b = next((o for o in fast[1:] if ctx.probe_placement(arena[a:a + nb], nb, out=arena[o:o + nb // 10 + 4096], reps=4) < 0.235), None)
    print(f"src at 0, class-2 window at {a >> 30} GiB, class-3 window at {None if b is None else b >> 30} GiB")
    tm = ctx.timer()
    for r, w in [(0, 8), (4, 8), (8, 8), (8, 4), (10, 1)]:
        n_items = 131072
        res = []
        for d1, d2 in ((a, a), (a, b)):
            p1, p2 = arena.data_ptr() + d1, arena.data_ptr() + d2
            for _ in range(3):
                assert fn(ctx.h, src.data_ptr(), p1, p2, n_items, r, w, 16, None, None) == 0
            tm.start(None)
            for _ in range(20):
                fn(ctx.h, src.data_ptr(), p1, p2, n_items, r, w, 16, None, None)
            tm.stop(None)
            ms = tm.elapsed_ms() / 20
            res.append((ms, n_items * (r + w) * 1024 / ms / 1e6))
        print(f"read{r}:write{w}  writes in one class {res[0][0]:.4f} ms {res[0][1]:.0f} GB/s | spread over two classes {res[1][0]:.4f} ms {res[1][1]:.0f} GB/s")
    # write pattern of the PCM store (64-byte segments at 128-byte stride, pairs of instructions) vs full 1 KiB per instruction
    for r, w in [(4, 8), (0, 8)]:
        for name, flag in (("1 KiB per store instruction", 0), ("64 B segments at 128 B stride", 0x80000000)):
            p1, p2 = arena.data_ptr() + a, arena.data_ptr() + (b if b is not None else a)
            for _ in range(3):
                fn(ctx.h, src.data_ptr(), p1, p2, 131072 | flag, r, w, 16, None, None)
            tm.start(None)
            for _ in range(20):
                fn(ctx.h, src.data_ptr(), p1, p2, 131072 | flag, r, w, 16, None, None)
            tm.stop(None)
            ms = tm.elapsed_ms() / 20
            print(f"read{r}:write{w} spread, {name}: {ms:.4f} ms {131072 * (r + w) * 1024 / ms / 1e6:.0f} GB/s")
    # reads spread over two classes (src + window a), records into the third (b): the meter's 10:1 mix
    if b is not None:
        for name, s2 in (("reads from one class", None), ("reads spread over two classes", arena.data_ptr() + a)):
            for _ in range(3):
                fn(ctx.h, src.data_ptr(), arena.data_ptr() + b, arena.data_ptr() + b, 131072, 10, 1, 16, None, s2)
            tm.start(None)
            for _ in range(20):
                fn(ctx.h, src.data_ptr(), arena.data_ptr() + b, arena.data_ptr() + b, 131072, 10, 1, 16, None, s2)
            tm.stop(None)
            ms = tm.elapsed_ms() / 20
            print(f"read10:write1  {name}: {ms:.4f} ms {131072 * 11 * 1024 / ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
