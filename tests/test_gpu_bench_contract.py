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
    for k in ("bound", "achieved", "peak", "unit", "frac", "frac_unassisted", "traffic"):
        assert k in r, k
    assert r["frac_unassisted"] > 0 and d["config"]["placement"]["headline_buffers"] == "abi"
    assert d["config"]["placement"]["io_alloc_report"]["placed"] == 0      # 0.3 MB of input: below the size where placement matters
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None                      # the committed PMC figure is for the full-size launch only
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    # the node aggregate is the oracle's over the same shard-invariant stream
    payload = orc.gen_uniform(F_ * C_ * 160).reshape(F_, C_, 160)
    st, agg = orc.decode_meter(payload, np.zeros((C_,), np.uint8), want_agg=True)
    assert d["aggregate"]["samples"] == F_ * C_ * 160 == int(agg["samples"])
    assert d["aggregate"]["node_peak"] == int(st["peak"].max())
    assert abs(d["aggregate"]["node_rms"] - float(np.sqrt(int(agg["sumsq"]) / int(agg["samples"])))) < 1e-3


@pytest.mark.parametrize("mode", ["store", "roundtrip", "rtp", "packets", "depayload", "encode"])
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
