"""world_size-2 and -8 gloo tests of the N>1 path's ONLY exchange step: the packed aggregate vector
(SURVEY 8e).  One sum all-reduce must yield the node-wide sums AND the max peak, because rank g
writes its local peak only into peak_slot[g].  The per-rank aggregates come from the oracle here
(no GPU in this container); the -m gpu suite checks that the kernels fill the same vector."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from igate4xsoftphonedsp_amd import capi
from igate4xsoftphonedsp_amd import dist as igdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as orc

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C_total, F_, n = 64, 5, 160
    lo, hi = igdist.channel_range(C_total, rank, world)
    # shard-invariant generation: this rank's channels of the global [F][C_total][160] array
    payload = np.stack([orc.gen_uniform((hi - lo) * n, first_byte=(f * C_total + lo) * n).reshape(hi - lo, n) for f in range(F_)])
    codec = np.where(np.arange(lo, hi) & 1, 8, 0).astype(np.uint8)
    _, agg = orc.decode_meter(payload, codec, want_agg=True, rank=rank)
    vec = torch.from_numpy(np.frombuffer(agg.tobytes(), dtype=np.int64).copy())
    node = igdist.allreduce_aggregate(vec)
    q.put((rank, node, int(agg["peak_slot"][rank])))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])                # 8 = the driver's scaling run: every peak slot, odd channel splits
def test_aggregate_allreduce(world):
    from oracle import oracle as orc

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda r: r[0])
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    # single-process truth over the whole array
    C_total, F_, n = 64, 5, 160
    payload = orc.gen_uniform(F_ * C_total * n).reshape(F_, C_total, n)
    codec = np.where(np.arange(C_total) & 1, 8, 0).astype(np.uint8)
    st, agg = orc.decode_meter(payload, codec, want_agg=True)
    for rank, node, local_peak in res:
        assert node["sumsq"] == int(agg["sumsq"]) and node["samples"] == C_total * F_ * n
        assert node["frames"] == C_total * F_ and node["byte_mean_sum"] == int(agg["byte_mean_sum"])
        assert node["peak"] == int(st["peak"].max()) == max(r[2] for r in res)
        assert abs(node["rms"] - np.sqrt(int(agg["sumsq"]) / (C_total * F_ * n))) < 1e-9
    assert all(r[1] == res[0][1] for r in res)          # every rank holds the same node-wide view


def test_channel_ranges_partition():
    for C_, G in ((65536, 8), (524288, 8), (10, 3), (7, 8), (4096, 1)):
        spans = [igdist.channel_range(C_, g, G) for g in range(G)]
        assert spans[0][0] == 0 and spans[-1][1] == C_
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert capi.AGG_WORDS == 112 and capi.AGG_LINE_WORDS == 16
