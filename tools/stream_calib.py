#!/usr/bin/env python3
"""Measured ceilings of this box's memory system for read : write mixes (bare load/store kernels, no compute):
prints one JSON object {mix: GB/s}.  Used to put the kernels' roofline fractions in context."""
import ctypes as C
import json
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
    n_items = 131072                                    # x up to 10 KiB = 1.34 GB, the headline batch
    nb = n_items * 10 * 1024
    # one arena; the write window goes where the 10:1 stream is fastest (another memory region than the read window)
    arena = torch.empty((2 * nb + (84 << 30),), dtype=torch.uint8, device="cuda")
    src = arena[:nb]
    ctx.gen_uniform(src, src.numel(), seed=1)
    torch.cuda.synchronize()
    best = None
    for k in range(8):
        off = nb + (k * 12 << 30)
        d = arena[off:off + nb]
        t = ctx.probe_placement(src, nb, out=d, reps=6)
        if best is None or t < best[0]:
            best = (t, k, d)
    dst = best[2] if "--same-region" not in sys.argv else arena[nb:2 * nb]
    print(f"write window at +{best[1] * 12} GiB (10:1 stream {best[0]:.4f} ms)" if "--same-region" not in sys.argv else "write window right behind the read window", file=sys.stderr)
    out = {}
    cases = [(0, 8, 16), (8, 8, 16), (8, 4, 16), (4, 8, 16), (10, 1, 16), (8, 1, 16), (8, 2, 16),
             (10, 1, 8), (10, 1, 4), (10, 1, 12), (20, 2, 16), (20, 2, 8), (5, 1, 16)]
    for r, w, wv in cases:
        n_items = 1342177280 // (max(r, 1) * 1024) if r >= 5 else 131072
        assert n_items * r * 1024 <= src.numel() and n_items * w * 1024 <= dst.numel()
        for _ in range(3):
            assert fn(ctx.h, src.data_ptr(), dst.data_ptr(), n_items, r, w, wv, None) == 0
        torch.cuda.synchronize()
        tm = ctx.timer()
        reps = 20
        tm.start(None)
        for _ in range(reps):
            fn(ctx.h, src.data_ptr(), dst.data_ptr(), n_items, r, w, wv, None)
        tm.stop(None)
        ms = tm.elapsed_ms() / reps
        out[f"read{r}:write{w}@{wv}waves"] = {"ms": round(ms, 4), "GBs": round(n_items * (r + w) * 1024 / (ms * 1e-3) / 1e9, 1),
                                    "read_GB": round(n_items * r * 1024 / 1e9, 3), "write_GB": round(n_items * w * 1024 / 1e9, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
