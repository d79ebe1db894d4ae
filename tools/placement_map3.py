#!/usr/bin/env python3
"""Bare read+record stream time with the payload window at several places of one very large allocation and the
record window swept over the whole allocation (8 GiB grid): which distances give which class."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    win = 65536 * 128 * 160
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 248
    big = torch.empty((gib << 30,), dtype=torch.uint8, device="cuda")
    grid = list(range(0, gib - 8, 8))
    print("record window offset (GiB): " + " ".join(f"{g:5d}" for g in grid))
    for a in (0, 64, 128, 192):
        if a + 8 > gib:
            break
        row = []
        for b in grid:
            bb = (b + 4) << 30
            row.append(ctx.probe_placement(big[(a << 30):(a << 30) + win], win, out=big[bb:bb + win // 10 + 4096], reps=5))
        print(f"payload at {a:3d} GiB:        " + " ".join("%.3f" % x for x in row), flush=True)


if __name__ == "__main__":
    main()
