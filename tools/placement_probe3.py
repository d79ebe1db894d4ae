#!/usr/bin/env python3
"""Real meter kernel timed over a matrix of payload-slab x record-buffer allocations in one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    C_, F_, n = 65536, 128, 160
    nbytes = C_ * F_ * n
    cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
    agg = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")
    pls = [torch.empty((nbytes,), dtype=torch.uint8, device="cuda") for _ in range(6)]
    sts = [torch.empty((F_ * C_ * 2 + (i << 18),), dtype=torch.int64, device="cuda") for i in range(5)]
    for p in pls:
        ctx.gen_uniform(p, nbytes, seed=3)
    torch.cuda.synchronize()
    tm = ctx.timer()
    print("bare probe per payload:", " ".join("%.4f" % ctx.probe_placement(p, nbytes, reps=10) for p in pls))
    for rnd in range(2):
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
            print(f"round {rnd} payload {i} ({p.data_ptr():#x}) x stats:", " ".join("%.4f" % x for x in row))


if __name__ == "__main__":
    main()
