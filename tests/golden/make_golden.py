#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/.  Run in the BUILD container:

    python tests/golden/make_golden.py

Sources of truth (none of them is our own oracle):
  * CPython stdlib ``audioop`` (independent ITU-T G.711 implementation; removed in
    Python 3.13, so the vectors are frozen here): full decode tables, exhaustive
    65 536-input encode tables (these define encoder variant G191).
  * The REAL reference recorder ``WavWriter.cpp`` compiled from /root/reference by
    oracle/Makefile into oracle/_ref/ (SURVEY.md 8c): a 4 ch x 50 frame recording.
  * SURVEY.md 8(c) known answers (SHA-256 of the decode tables, spot values).

Inputs fed to the reference recorder come from the oracle's D-speech generator;
they are stored in the fixture too, so the fixture is self-contained data.
"""
import audioop
import hashlib
import json
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402  (test infrastructure)


def main():
    allcodes = bytes(range(256))
    dec_u = np.frombuffer(audioop.ulaw2lin(allcodes, 2), dtype="<i2")
    dec_a = np.frombuffer(audioop.alaw2lin(allcodes, 2), dtype="<i2")
    pcm_all = np.arange(-32768, 32768, dtype="<i2").tobytes()
    enc_u = np.frombuffer(audioop.lin2ulaw(pcm_all, 2), dtype=np.uint8)
    enc_a = np.frombuffer(audioop.lin2alaw(pcm_all, 2), dtype=np.uint8)
    np.savez_compressed(
        os.path.join(HERE, "g711_audioop.npz"),
        ulaw_decode=dec_u, alaw_decode=dec_a, ulaw_encode_g191=enc_u, alaw_encode_g191=enc_a,
    )

    kat = {
        "source": "SURVEY.md 8(c); tables hashed as 256 x int16 little-endian",
        "ulaw_decode_sha256": hashlib.sha256(dec_u.tobytes()).hexdigest(),
        "alaw_decode_sha256": hashlib.sha256(dec_a.tobytes()).hexdigest(),
        "ulaw_spot": {"0x00": -32124, "0x01": -31100, "0x7F": 0, "0x80": 32124, "0xFE": 8, "0xFF": 0},
        "alaw_spot": {"0x00": -5504, "0x2A": -32256, "0x55": -8, "0x80": 5504, "0xAA": 32256, "0xD5": 8, "0xFF": 848},
        "ulaw_max_abs": 32124,
        "alaw_max_abs": 32256,
        "closed_set_exceptions": {"ulaw": {"0x7F": "0xFF"}, "alaw": {}},
    }
    assert kat["ulaw_decode_sha256"] == "3dab54339e520bb2c924826e3b72a917a2b612e9fd12fc867500f1d983a75827"
    assert kat["alaw_decode_sha256"] == "e04788d110e58ff8c70c93b8480190d973e3b67876b6119abbaec766cc75c174"
    with open(os.path.join(HERE, "g711_kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1, sort_keys=True)

    # --- config #1: 4 ch x 50 frames (1 s) mu-law D-speech through the REAL WavWriter
    C_, F_, n = 4, 50, 160
    codec = np.zeros((C_,), dtype=np.uint8)
    payload = orc.gen_speech(C_, F_, n, codec, variant=orc.ENC_G191)  # encoder pinned by audioop
    wavs = []
    if not orc.ref_wavwriter_available():
        raise SystemExit("oracle/_ref/libref_wavwriter.so missing: run `make -C oracle` with /root/reference mounted")
    with tempfile.TemporaryDirectory() as td:
        for c in range(C_):  # file-scope globals => one recorder at a time (WavWriter.cpp:5-15)
            wavs.append(np.frombuffer(orc.ref_wav_record(td, payload[:, c, :]), dtype=np.uint8))
    # audioop-derived expected levels for the same frames (independent of our oracle)
    rms_floor = np.zeros((F_, C_), dtype=np.int64)
    peak = np.zeros((F_, C_), dtype=np.int64)
    sumsq = np.zeros((F_, C_), dtype=np.uint64)
    bmean = np.zeros((F_, C_), dtype=np.uint8)
    for f in range(F_):
        for c in range(C_):
            lin = audioop.ulaw2lin(payload[f, c].tobytes(), 2)
            rms_floor[f, c] = audioop.rms(lin, 2)
            peak[f, c] = audioop.max(lin, 2)
            x = np.frombuffer(lin, dtype="<i2").astype(np.int64)
            sumsq[f, c] = int((x * x).sum())
            bmean[f, c] = sum(payload[f, c].tolist()) // n
    np.savez_compressed(
        os.path.join(HERE, "config1_4ch_50f.npz"),
        payload=payload, codec=codec, wav0=wavs[0], wav1=wavs[1], wav2=wavs[2], wav3=wavs[3],
        audioop_rms_floor=rms_floor, audioop_peak=peak, sumsq=sumsq, byte_mean=bmean,
    )
    meta = {
        "wav_sha256": [hashlib.sha256(w.tobytes()).hexdigest() for w in wavs],
        "wav_len": [int(w.size) for w in wavs],
        "payload_sha256": hashlib.sha256(payload.tobytes()).hexdigest(),
        "generator": "oracle.gen_speech(C=4,F=50,n=160,codec=0,seed=0x20241218,variant=G191)",
    }
    with open(os.path.join(HERE, "config1_4ch_50f.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)

    # --- percent scale from the REAL AudioMeter (audiometer.cpp:16-34 through its FIFO)
    if not orc.ref_audiometer_available():
        raise SystemExit("oracle/_ref/libref_audiometer.so missing: run `make -C oracle` (needs Qt5 moc under /opt/conda)")
    levels = [0, 1, 299, 300, 301, 599, 600, 14999, 15000, 29999, 30000, 30299, 30300, 32124, 32256, 32767, 65535, 123456, -1, -299, -300, -30000]
    levels += list(range(0, 33000, 997))
    with open(os.path.join(HERE, "audiometer_percent.json"), "w") as fh:
        json.dump({"source": "AudioMeter::getAudioLevel() of /root/reference/audiometer.cpp (real object, via /tmp/capturefifo<card>)",
                   "levels": levels, "percent": orc.ref_audiometer_percent(levels)}, fh)

    # --- splitmix64 known answers (public reference values for seed 0 stream: 1234567 variant below is ours)
    sm = {"splitmix64(0)": str(orc.lib().orc_splitmix64(0)), "first16_uniform_seed": orc.gen_uniform(16).tolist()}
    with open(os.path.join(HERE, "prng.json"), "w") as fh:
        json.dump(sm, fh, indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)
    for fn in sorted(os.listdir(HERE)):
        print("  ", fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
