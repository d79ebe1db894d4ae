"""-m "not gpu": AddressSanitizer + UBSan over the CPU-side code — the oracle (the checker every parity test trusts)
and the C++ host mirror's GPU-free entry points — driven by tests/san/san_driver.cpp with random and hostile inputs
(lengths past n, sizes past the slot, runts, negative sizes, zero-capacity buffers).  GPU sanitizers are not available
on the pool, so this is the sanitizer coverage the project has."""
import os
import shutil
import subprocess

import pytest

from igate4xsoftphonedsp_amd import build as igbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_asan():
    try:
        out = subprocess.run(["g++", "-fsanitize=address,undefined", "-x", "c++", "-", "-o", os.devnull], input=b"int main(){return 0;}",
                             capture_output=True, timeout=60)
        return out.returncode == 0
    except Exception:
        return False


@pytest.mark.skipif(shutil.which("g++") is None or not _has_asan(), reason="g++ with libasan/libubsan not available")
def test_oracle_and_host_mirror_under_asan_ubsan(tmp_path):
    igbuild.build()
    exe = tmp_path / "san_driver"
    pkg = os.path.join(ROOT, "igate4xsoftphonedsp_amd")
    cmd = ["g++", "-std=c++11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-funsigned-char", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(pkg, "host"), "-I", os.path.join(ROOT, "oracle"),
           os.path.join(ROOT, "tests", "san", "san_driver.cpp"), os.path.join(pkg, "host", "igdsp_host.cpp"),
           "-x", "c", os.path.join(ROOT, "oracle", "igdsp_oracle.c"), "-x", "none",
           "-L", pkg, "-ligdsp", f"-Wl,-rpath,{pkg}", "-lpthread", "-lm", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "oracle pass ok" in r.stdout and "host pass ok" in r.stdout
