#!/usr/bin/env python3
"""Map of the read+record stream rate over ONE large device allocation: probe 1.34 GB windows at 256 MiB steps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    win = 65536 * 128 * 160
    total = int(sys.argv[1]) << 30 if len(sys.argv) > 1 else 48 << 30
    step = 256 << 20
    big = torch.empty((total,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    print(f"base {big.data_ptr():#x} total {total >> 30} GiB")
    line = []
    for off in range(0, total - win, step):
        ms = ctx.probe_placement(big[off:off + win], win, reps=6)
        line.append(ms)
        if len(line) == 16:
            print(f"{(off - 15 * step) / 2**30:7.2f} GiB: " + " ".join("%.3f" % x for x in line), flush=True)
            line = []
    if line:
        print("tail: " + " ".join("%.3f" % x for x in line))


if __name__ == "__main__":
    main()
