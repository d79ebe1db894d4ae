#!/usr/bin/env python3
"""The three classes of device memory seen from ONE large hipMalloc (round-1 view; the product's own view, chunk by chunk, is
tools/io_survey.py).  Sub-commands:
  read  [GiB]   pure read-stream time of 1.34 GB windows at 256 MiB steps (the read rate is the same everywhere)
  grid          bare read + record stream over an 8 GiB grid of (payload window, record window) positions inside 80 GiB
  sweep [GiB]   payload window at 0 / 64 / 128 / 192 GiB, record window swept over the whole allocation (default 248 GiB)
  kernel        the real meter kernel over a matrix of separately allocated payload slabs x record buffers"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402

WIN = 65536 * 128 * 160


def read_map(ctx, gib):
    total, step = gib << 30, 256 << 20
    big = torch.empty((total,), dtype=torch.uint8, device="cuda")
    print(f"base {big.data_ptr():#x} total {gib} GiB")
    line = []
    for off in range(0, total - WIN, step):
        line.append(ctx.probe_placement(big[off:off + WIN], WIN, reps=6))
        if len(line) == 16:
            print(f"{(off - 15 * step) / 2**30:7.2f} GiB: " + " ".join("%.3f" % x for x in line), flush=True)
            line = []
    if line:
        print("tail: " + " ".join("%.3f" % x for x in line))


def grid(ctx):
    big = torch.empty((80 << 30,), dtype=torch.uint8, device="cuda")
    cells = [g << 30 for g in range(0, 80, 8)]
    print("rows: payload window offset; columns: record window offset (GiB): " + " ".join(f"{g >> 30:5d}" for g in cells))
    for a in cells:
        row = [ctx.probe_placement(big[a:a + WIN], WIN, out=big[b + (4 << 30):b + (4 << 30) + WIN // 10 + 4096], reps=6) for b in cells]
        print(f"{a >> 30:3d} GiB: " + " ".join("%.3f" % x for x in row), flush=True)


def sweep(ctx, gib):
    big = torch.empty((gib << 30,), dtype=torch.uint8, device="cuda")
    cells = list(range(0, gib - 8, 8))
    print("record window offset (GiB): " + " ".join(f"{g:5d}" for g in cells))
    for a in (0, 64, 128, 192):
        if a + 8 > gib:
            break
        row = [ctx.probe_placement(big[(a << 30):(a << 30) + WIN], WIN, out=big[(b + 4) << 30:((b + 4) << 30) + WIN // 10 + 4096], reps=5) for b in cells]
        print(f"payload at {a:3d} GiB:        " + " ".join("%.3f" % x for x in row), flush=True)


def kernel(ctx):
    C_, F_, n = 65536, 128, 160
    cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
    agg = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")
    pls = [torch.empty((WIN,), dtype=torch.uint8, device="cuda") for _ in range(6)]
    sts = [torch.empty((F_ * C_ * 2 + (i << 18),), dtype=torch.int64, device="cuda") for i in range(5)]
    for p in pls:
        ctx.gen_uniform(p, WIN, seed=3)
    torch.cuda.synchronize()
    tm = ctx.timer()
    for i, p in enumerate(pls):
        row = []
        for st in sts:
            for _ in range(2):
                ctx.decode_meter(p, cd, C_, F_, n, st, agg=agg)
            tm.start(None)
            for _ in range(10):
                ctx.decode_meter(p, cd, C_, F_, n, st, agg=agg)
            tm.stop(None)
            row.append(tm.elapsed_ms() / 10)
        print(f"payload {i} ({p.data_ptr():#x}) x record buffers:", " ".join("%.4f" % x for x in row), flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "sweep"
    arg = int(sys.argv[2]) if len(sys.argv) > 2 else None
    ctx = capi.Context(device=0, max_channels=64)
    {"read": lambda: read_map(ctx, arg or 48), "grid": lambda: grid(ctx), "sweep": lambda: sweep(ctx, arg or 248), "kernel": lambda: kernel(ctx)}[cmd]()
