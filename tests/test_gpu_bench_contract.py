"""-m gpu: bench.py prints ONE JSON line with the contract's keys (driver contract + roofline + cpu_baseline), at a
reduced channel count so it finishes in seconds; the aggregate it reports equals the oracle's on the same generator."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--prewarm-ms", "0",
           "--cpu-seconds", "0.5"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_json_contract_small(orc):
    C_, F_ = 256, 8
    d = _run(["--channels", str(C_), "--frames", str(F_)])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "frac_plain_warm", "frac_plain_cold", "traffic", "median_ms", "min_ms", "max_ms"):
        assert k in r, k
    assert r["frac_plain_warm"] > 0 and r["frac_plain_cold"] > 0 and d["config"]["placement"]["headline_buffers"] == "abi"
    assert r["min_ms"] <= r["median_ms"] <= r["max_ms"] and r["median_launches"] >= 20
    assert "placement_setup_ms" in d and d["placement_setup_ms"] >= 0
    assert d["config"]["placement"]["io_alloc_report"]["placed"] == 0      # 0.3 MB of input: below the size where placement matters
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None                      # the committed PMC figure is for the full-size launch only
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["batch_loop_value"] > 0
    assert "ONE FRAME PER CALL" in cb["sample"]          # B1 as BASELINE.md section 2 words it: through the single-frame entry
    # the node aggregate is the oracle's over the same shard-invariant stream
    payload = orc.gen_uniform(F_ * C_ * 160).reshape(F_, C_, 160)
    st, agg = orc.decode_meter(payload, np.zeros((C_,), np.uint8), want_agg=True)
    assert d["aggregate"]["samples"] == F_ * C_ * 160 == int(agg["samples"])
    assert d["aggregate"]["node_peak"] == int(st["peak"].max())
    assert abs(d["aggregate"]["node_rms"] - float(np.sqrt(int(agg["sumsq"]) / int(agg["samples"])))) < 1e-3


@pytest.mark.parametrize("mode", ["store", "roundtrip", "rtp", "packets", "window", "depayload", "encode"])
def test_bench_secondary_modes_run(mode):
    d = _run(["--channels", "1024", "--frames", "16", "--mode", mode, "--no-cpu-baseline"])
    assert d["value"] > 0 and d["roofline"]["achieved"] > 0 and d["cpu_baseline"] is None


def test_bench_force_collective_runs_rccl_on_one_gpu():
    """The N > 1 step sequence (kernel -> event -> side-stream all_reduce(int64[112]) over nccl = RCCL -> next launch) on the one
    GPU of the box, in bench.py's own process: a 1-rank group is brought up before any other GPU work."""
    d = _run(["--channels", "4096", "--frames", "16", "--force-collective", "--no-cpu-baseline"])
    c = d["collective"]
    assert c["world"] == 1 and c["ms_per_step_with_allreduce"] > 0 and c["ms_per_step_without"] > 0
    assert d["aggregate"]["samples"] == 4096 * 16 * 160


def _run_ranks(nproc, extra):
    """The driver's N > 1 command line (python -m torch.distributed.run ... bench.py --gpus N), with the one deviation a
    one-GPU box forces: both ranks use cuda:0 and the 896-byte all-reduce travels over gloo instead of RCCL."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--prewarm-ms", "0", "--backend", "gloo", "--one-gpu-rehearsal"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 alone prints
    return json.loads(lines[0])


@pytest.mark.parametrize("strong", [False, True])
def test_bench_two_ranks_keep_the_world_gt_1_branch_alive(orc, strong):
    """bench.py's world > 1 branch (process group, channel sharding by rank, per-launch event + side-stream all-reduce, barriers,
    max-over-ranks time, rank-0 JSON) started exactly as the driver starts it.  No scaling figure is read off this run — two
    ranks share one GPU; what is checked is that the branch runs and that the node-wide aggregate equals the oracle's over BOTH
    shards of the shard-invariant stream (weak: 256 ch per rank; strong: 512 ch split over the ranks)."""
    C_rank, F_, world = 256, 8, 2
    extra = ["--frames", str(F_)] + (["--total-channels", str(C_rank * world)] if strong else ["--channels", str(C_rank)])
    d = _run_ranks(world, extra)
    assert d["n_gpus"] == world and d["steps"] == 3 and d["scaling"] == ("strong" if strong else "weak")
    assert d["config"]["channels_per_gpu"] == C_rank and d["config"]["channels_total"] == C_rank * world
    assert d["cpu_baseline"] is None and d["value"] > 0 and d["roofline"]["kernel_avg_ms"] > 0
    C_total = C_rank * world
    payload = orc.gen_uniform(F_ * C_total * 160).reshape(F_, C_total, 160)
    st, agg = orc.decode_meter(payload, np.zeros((C_total,), np.uint8), want_agg=True)
    assert d["aggregate"]["samples"] == F_ * C_total * 160 == int(agg["samples"])
    assert d["aggregate"]["node_peak"] == int(st["peak"].max())
    assert abs(d["aggregate"]["node_rms"] - float(np.sqrt(int(agg["sumsq"]) / int(agg["samples"])))) < 1e-3
