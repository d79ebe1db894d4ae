#!/usr/bin/env python3
"""A/B inside ONE process (same clocks, same buffers — separate processes on the same box differ by up to 10 %):
the headline launch with and without the launch aggregate, interleaved groups, HIP-event timed.

History (r01): with all seven counters of igdsp_aggregate on one 128-byte line the aggregate cost 8-10 us per
0.25 ms launch (the per-block atomics of the 256 blocks serialise on that line); a ticketed scheme (per-block scratch
lines + last-block fold) cost 7 us (three dependent round trips); one line per counter costs 3-4 us, of which ~1 us
is the in-loop accumulation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from igate4xsoftphonedsp_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(device=0, max_channels=64)
    C_, F_, n = 65536, 128, 160
    pl = torch.empty((F_ * C_ * n,), dtype=torch.uint8, device="cuda")
    ctx.gen_uniform(pl, pl.numel(), seed=0x20241218)
    cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
    st = torch.zeros((F_ * C_ * 2,), dtype=torch.int64, device="cuda")
    agg = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    res = {"agg": [], "noagg": []}
    tm = ctx.timer()
    reps, warm, timed = 8, 3, 25
    for _ in range(reps):
        for name in res:
            a = agg if name == "agg" else None
            for _ in range(warm):
                ctx.decode_meter(pl, cd, C_, F_, n, st, agg=a)
            tm.start(None)
            for _ in range(timed):
                ctx.decode_meter(pl, cd, C_, F_, n, st, agg=a)
            tm.stop(None)
            res[name].append(tm.elapsed_ms() / timed)
    for k, v in res.items():
        print(k, "median %.4f ms" % float(np.median(v)), " ".join("%.4f" % x for x in v))
    torch.cuda.synchronize()
    frames = int(agg.cpu().numpy()[2 * capi.AGG_LINE_WORDS])
    assert frames == (warm + timed) * reps * C_ * F_, frames


if __name__ == "__main__":
    main()
