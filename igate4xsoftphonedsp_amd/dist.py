"""Multi-GPU sharding helpers.  The path shards trivially by channel (no cross-channel term exists,
roip_ed137.cpp:6564-6585): rank g owns the contiguous range [g*C/G, (g+1)*C/G) of channels, its own
payload slab and hold state.  The ONLY collective is one sum all-reduce of the 112-word (896-byte: seven counters, one
128-byte line each) launch aggregate (igdsp_aggregate, include/igdsp.h) — RCCL over xGMI on GPUs (backend "nccl"), gloo in
CPU tests.  Hardware evidence so far: a 1-rank RCCL group on one MI355X (bench.py --force-collective, tests/test_gpu_bench_contract.py);
N > 1 on real xGMI is only ever run by the driver's 8-GPU scaling bench.  """
from __future__ import annotations

import math

import torch
import torch.distributed as dist

from . import capi


def channel_range(n_channels: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous channel range of `rank` (balanced to within one channel)."""
    return (n_channels * rank) // world, (n_channels * (rank + 1)) // world


def allreduce_aggregate(vec: torch.Tensor, async_op: bool = False):
    """Sum-all-reduce the packed aggregate in place (int64[capi.AGG_WORDS]; device tensor under nccl) and return the
    node-wide view.  Rank g's peak sits alone in peak_slot[g], so the SUM also delivers every rank's
    peak and the max is taken locally: sums and max in one collective of 896 bytes (each counter sits on its own 128-byte line, zero padding between)."""
    assert vec.dtype == torch.int64 and vec.numel() == capi.AGG_WORDS
    work = None
    if dist.is_initialized() and dist.get_world_size() > 1:
        work = dist.all_reduce(vec, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            return work
    return node_view(vec)


def node_view(vec: torch.Tensor) -> dict:
    v = [int(x) & 0xFFFFFFFFFFFFFFFF for x in vec.cpu().tolist()]
    L = capi.AGG_LINE_WORDS                                   # counter k lives in word k * L
    samples = max(v[1 * L], 1)
    return {
        "sumsq": v[0], "samples": v[1 * L], "frames": v[2 * L], "n_silent": v[3 * L], "n_clipped": v[4 * L],
        "byte_mean_sum": v[5 * L], "peak": max(v[6 * L:6 * L + capi.AGG_MAX_RANKS]), "rms": math.sqrt(v[0] / samples),
    }
