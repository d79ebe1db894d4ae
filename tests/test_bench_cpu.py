"""-m "not gpu": bench.py's command line (no GPU needed): --help lists the contract's flags, and the two refusals
(no GPU; --gpus N without a distributed launch) are loud and specific."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, cwd=ROOT)


def test_help_lists_contract_flags():
    r = _bench("--help")
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--channels", "--frames", "--total-channels", "--placement", "--force-collective", "--prewarm-ms"):
        assert flag in r.stdout, flag


def test_refuses_multi_gpu_without_launcher_and_cpu_only():
    r = _bench("--gpus", "2")
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)
    import torch
    if not torch.cuda.is_available():
        r = _bench("--steps", "1")
        assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
