"""CPU checks of the C++ host mirror: the WavWriter-compatible recorder reproduces the REAL
reference recorder's bytes (golden fixture written by /root/reference/WavWriter.cpp), and the
library loads without a GPU."""
import hashlib
import json
import os

import numpy as np

from tests import host_util as hu


def test_recorder_bytes_equal_real_wavwriter(golden_dir, tmp_path):
    L = hu.load()
    g = np.load(os.path.join(golden_dir, "config1_4ch_50f.npz"))
    with open(os.path.join(golden_dir, "config1_4ch_50f.json")) as fh:
        meta = json.load(fh)
    pkt = bytes(12)
    for c in range(4):
        path = str(tmp_path / f"rec{c}.wav").encode()
        w = L.igdsp_wav_start(path, 8000)
        assert w
        for f in range(50):
            pl = g["payload"][f, c].tobytes()
            assert L.igdsp_wav_writeRTPWav(w, pkt, pl, 12, len(pl)) == 0
        assert L.igdsp_wav_stop(w) == 0
        mine = open(path, "rb").read()
        assert len(mine) == meta["wav_len"][c]
        assert hashlib.sha256(mine).hexdigest() == meta["wav_sha256"][c]
        assert mine == g[f"wav{c}"].tobytes()


def test_recorder_rejects_bad_args(tmp_path):
    L = hu.load()
    assert L.igdsp_wav_writeRTPWav(None, None, None, 0, 0) == -22
    assert not L.igdsp_wav_start(str(tmp_path / "nodir" / "x.wav").encode(), 8000)
    assert L.igdsp_wav_stop(None) == -22


def test_host_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present")
    L = hu.load()
    assert not L.igdsp_host_create(0, 4)          # NULL: no device, no CPU metering path
    assert L.igdsp_host_tick(None, None) == -22
