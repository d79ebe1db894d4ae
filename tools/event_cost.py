#!/usr/bin/env python3
"""Does one event record per launch on the launch stream (what the N > 1 path needs for its side-stream all-reduce)
slow the back-to-back launches down?  One process, interleaved groups."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    C_, F_, n = 65536, 128, 160
    nbytes = C_ * F_ * n
    main_s, comm_s = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.set_stream(main_s)
    hs = main_s.cuda_stream
    pl = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    ctx.gen_uniform(pl, nbytes, seed=1, stream=hs)
    cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
    st = torch.zeros((F_ * C_ * 2,), dtype=torch.int64, device="cuda")
    aggs = torch.zeros((64, capi.AGG_WORDS), dtype=torch.int64, device="cuda")
    evs = [torch.cuda.Event() for _ in range(64)]
    tm = ctx.timer()
    res = {"plain": [], "event": [], "event+side-kernel": []}
    for _ in range(6):
        for name in res:
            for _ in range(5):
                ctx.decode_meter(pl, cd, C_, F_, n, st, agg=aggs[0], stream=hs)
            tm.start(hs)
            for i in range(50):
                ctx.decode_meter(pl, cd, C_, F_, n, st, agg=aggs[i], stream=hs)
                if name != "plain":
                    evs[i].record(main_s)
                    if name == "event+side-kernel":
                        with torch.cuda.stream(comm_s):
                            comm_s.wait_event(evs[i])
                            aggs[i].add_(0)                      # stands in for the all-reduce kernel
            tm.stop(hs)
            res[name].append(tm.elapsed_ms() / 50)
            torch.cuda.synchronize()
    for k, v in res.items():
        print(f"{k:18s} median {float(np.median(v)):.4f} ms  " + " ".join("%.4f" % x for x in v))


if __name__ == "__main__":
    main()
