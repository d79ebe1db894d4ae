"""Pins the CPU oracle (oracle/) against every external vector we have:
SURVEY.md 8(c) KATs, the frozen CPython-audioop tables, the REAL reference
WavWriter output, and an independent numpy restatement.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def aud(golden_dir):
    return np.load(os.path.join(golden_dir, "g711_audioop.npz"))


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "g711_kat.json")) as fh:
        return json.load(fh)


def test_decode_tables_match_audioop_and_sha(orc, aud, kat):
    tu, ta = orc.decode_table(0), orc.decode_table(8)
    assert np.array_equal(tu, aud["ulaw_decode"])
    assert np.array_equal(ta, aud["alaw_decode"])
    assert hashlib.sha256(tu.astype("<i2").tobytes()).hexdigest() == kat["ulaw_decode_sha256"]
    assert hashlib.sha256(ta.astype("<i2").tobytes()).hexdigest() == kat["alaw_decode_sha256"]


def test_decode_spot_values(orc, kat):
    tu, ta = orc.decode_table(0), orc.decode_table(8)
    for k, v in kat["ulaw_spot"].items():
        assert tu[int(k, 16)] == v
    for k, v in kat["alaw_spot"].items():
        assert ta[int(k, 16)] == v
    assert np.abs(tu.astype(int)).max() == kat["ulaw_max_abs"]
    assert np.abs(ta.astype(int)).max() == kat["alaw_max_abs"]


def test_numpy_restatement_agrees(orc):
    codes = np.arange(256, dtype=np.uint8)
    assert np.array_equal(orc.np_ulaw2lin(codes), orc.decode_table(0))
    assert np.array_equal(orc.np_alaw2lin(codes), orc.decode_table(8))


def test_encode_g191_exhaustive_vs_audioop(orc, aud):
    assert np.array_equal(orc.encode_table(0, orc.ENC_G191), aud["ulaw_encode_g191"])
    assert np.array_equal(orc.encode_table(8, orc.ENC_G191), aud["alaw_encode_g191"])


@pytest.mark.parametrize("variant", [0, 1])
def test_encode_closed_set(orc, kat, variant):
    """enc(dec(c)) == c for every code in both encoder lineages, except mu-law 0x7F -> 0xFF
    (negative zero folds onto positive zero)."""
    for pt, law in ((0, "ulaw"), (8, "alaw")):
        tab = orc.decode_table(pt)
        enc = orc.encode_table(pt, variant)
        back = enc[tab.astype(np.int32) + 32768]
        exp = np.arange(256, dtype=np.uint8)
        for k, v in kat["closed_set_exceptions"][law].items():
            exp[int(k, 16)] = int(v, 16)
        assert np.array_equal(back, exp), (law, variant)


@pytest.mark.parametrize("variant", [0, 1])
def test_encode_monotone_and_decision_values(orc, variant):
    """ITU-T G.711 properties every conforming encoder has: the decoded value of
    enc(x) is monotone non-decreasing in x, sign-consistent, and |dec(enc(x)) - x|
    never exceeds half the local step (+1 LSB slack for the two rounding lineages)."""
    x = np.arange(-32768, 32768, dtype=np.int64)
    for pt in (0, 8):
        tab = orc.decode_table(pt).astype(np.int64)
        y = tab[orc.encode_table(pt, variant)]
        assert np.all(np.diff(y) >= 0)
        lim = 32124 if pt == 0 else 32256
        inr = np.abs(x) <= lim
        err = np.abs(y - x)[inr]
        step = np.maximum(np.abs(x[inr]) // 16, 8 if pt == 0 else 16)   # segment step bound
        assert np.all(err <= step + 8), (pt, variant, err.max())


def test_encoder_variants_differ_only_on_negatives_or_clip(orc):
    """Documents where the two lineages differ (SURVEY.md section 7 'hard parts')."""
    x = np.arange(-32768, 32768, dtype=np.int64)
    for pt in (0, 8):
        a, b = orc.encode_table(pt, 0), orc.encode_table(pt, 1)
        diff = x[a != b]
        assert diff.size > 0
        if pt == 8:
            assert np.all(diff < 0)        # A-law: only the '-pcm-8' vs '-pcm-1' rounding of negatives
        else:                              # mu-law: lineages differ by at most one quantisation step
            tab = orc.decode_table(0).astype(np.int64)
            assert np.abs(tab[a] - tab[b]).max() <= 1024


def test_byte_mean_restates_reference_loop(orc):
    rng = np.random.default_rng(1)
    for n in (160, 164, 24, 1, 256):
        buf = rng.integers(0, 256, size=n, dtype=np.uint8)
        assert orc.byte_mean(buf) == int(buf.astype(np.int64).sum()) // n      # aarch64: unsigned char
    # the x86 (signed char) build of the same loop truncates differently: document it
    buf = np.full(160, 0xFF, dtype=np.uint8)
    assert orc.byte_mean(buf) == 255
    assert orc.byte_mean(buf, signed_char=True) == 255          # (uint8_t)(-1)
    buf = np.full(160, 0x80, dtype=np.uint8)
    assert orc.byte_mean(buf) == 128 and orc.byte_mean(buf, signed_char=True) == 128
    buf = np.array([0x80, 0x7F] * 80, dtype=np.uint8)
    assert orc.byte_mean(buf) == 127 and orc.byte_mean(buf, signed_char=True) == 0


def test_percent_scale(orc):
    # audiometer.cpp:30-31 int(float(v*100.0/30000.0))
    for v, p in ((0, 0), (299, 0), (300, 1), (15000, 50), (30000, 100), (32767, 109), (-300, -1)):
        assert orc.percent(v) == p


def test_percent_scale_against_real_audiometer(orc, golden_dir):
    """The fixture holds what the REAL reference AudioMeter emitted (object code built from
    /root/reference/audiometer.cpp); the restated formula must agree on every level, live too when oracle/_ref exists."""
    with open(os.path.join(golden_dir, "audiometer_percent.json")) as fh:
        g = json.load(fh)
    assert len(g["levels"]) == len(g["percent"]) > 50
    for v, p in zip(g["levels"], g["percent"]):
        assert orc.percent(v) == p, v
    if orc.ref_audiometer_available():
        live = [7, 300, 29999, 31000, -450]
        assert orc.ref_audiometer_percent(live) == [orc.percent(v) for v in live]


def test_config1_against_real_wavwriter_and_audioop(orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "config1_4ch_50f.npz"))
    with open(os.path.join(golden_dir, "config1_4ch_50f.json")) as fh:
        meta = json.load(fh)
    payload, codec = g["payload"], g["codec"]
    assert hashlib.sha256(payload.tobytes()).hexdigest() == meta["payload_sha256"]
    # the oracle's generator still reproduces the committed payload
    assert np.array_equal(orc.gen_speech(4, 50, 160, codec, variant=orc.ENC_G191), payload)
    stats = orc.decode_meter(payload, codec)
    assert np.array_equal(stats["sumsq"], g["sumsq"])
    assert np.array_equal(stats["peak"].astype(np.int64), g["audioop_peak"])
    assert np.array_equal(stats["byte_mean"], g["byte_mean"])
    # audioop.rms is floor(sqrt(mean)): a sanity bound on the float RMS
    assert np.all(np.floor(stats["rms"].astype(np.float64) + 1e-3).astype(np.int64) >= g["audioop_rms_floor"])
    assert np.all(stats["rms"].astype(np.float64) < g["audioop_rms_floor"] + 1.0 + 1e-3)
    # recorder bytes: restated header + [b,0] expansion == what the REAL WavWriter wrote
    for c in range(4):
        body = orc.wav_expand(payload[:, c, :])
        mine = np.concatenate([np.frombuffer(orc.wav_header(8000, body.size), dtype=np.uint8), body])
        ref = g[f"wav{c}"]
        assert mine.size == ref.size == meta["wav_len"][c]
        assert np.array_equal(mine, ref)
        assert hashlib.sha256(mine.tobytes()).hexdigest() == meta["wav_sha256"][c]


def test_real_wavwriter_live_if_built(orc, tmp_path):
    """When oracle/_ref is present (it travels to the GPU box as a built .so), run the
    REAL reference recorder again and compare with the restatement on fresh input."""
    if not orc.ref_wavwriter_available():
        pytest.skip("oracle/_ref not built (reference not mounted)")
    payload = orc.gen_uniform(7 * 160, seed=99).reshape(7, 160)
    ref = np.frombuffer(orc.ref_wav_record(str(tmp_path), payload), dtype=np.uint8)
    body = orc.wav_expand(payload)
    mine = np.concatenate([np.frombuffer(orc.wav_header(8000, body.size), dtype=np.uint8), body])
    assert np.array_equal(mine, ref)


def test_prng_fixture(orc, golden_dir):
    with open(os.path.join(golden_dir, "prng.json")) as fh:
        p = json.load(fh)
    assert str(orc.lib().orc_splitmix64(0)) == p["splitmix64(0)"] == "16294208416658607535"
    assert orc.gen_uniform(16).tolist() == p["first16_uniform_seed"]
    # shard invariance: any window of the stream equals the same window generated alone
    whole = orc.gen_uniform(4096)
    assert np.array_equal(orc.gen_uniform(1000, first_byte=777), whole[777:1777])


def test_frame_flags_and_edges(orc):
    n = 160
    mk = lambda b: np.full((1, 1, n), b, dtype=np.uint8)
    s = orc.decode_meter(mk(0xFF), [0])[0, 0]
    assert s["sumsq"] == 0 and s["peak"] == 0 and s["flags"] & orc.FLAG_SILENT and s["byte_mean"] == 255
    s = orc.decode_meter(mk(0xD5), [8])[0, 0]
    assert s["peak"] == 8 and s["sumsq"] == 64 * n and s["flags"] == (orc.FLAG_SILENT | orc.FLAG_PROBE_D5)
    s = orc.decode_meter(mk(0x00), [0])[0, 0]
    assert s["peak"] == 32124 and s["flags"] & orc.FLAG_CLIPPED and s["sumsq"] == 32124 ** 2 * n
    s = orc.decode_meter(mk(0x2A), [8])[0, 0]
    assert s["peak"] == 32256 and s["sumsq"] == 32256 ** 2 * n and abs(s["rms"] - 32256.0) < 1e-2
    # ragged / empty
    pl = orc.gen_uniform(3 * 2 * n).reshape(3, 2, n)
    ln = np.array([[160, 0], [24, 159], [1, 160]], dtype=np.uint16)
    st, pcm = orc.decode_meter(pl, [0, 8], length=ln, want_pcm=True)
    assert st[0, 1]["flags"] == orc.FLAG_EMPTY and st[0, 1]["sumsq"] == 0
    assert st[1, 0]["byte_mean"] == int(pl[1, 0, :24].astype(int).sum()) // 24
    assert np.all(pcm[1, 0, 24:] == 0)


def test_hold_semantics_follow_ptt_logger(orc):
    """keeplogAudioLevel (Functions.cpp:2126-2145): count/sum/max/min; init max 0, min 255;
    the reference's uint16 OutgoingRTPSum is our exact level_sum mod 65536."""
    C_, F_, n = 3, 300, 160
    pl = np.full((F_, C_, n), 0xFF, dtype=np.uint8)       # byte_mean 255 every frame
    st = orc.decode_meter(pl, [0, 0, 8])
    h = orc.hold_new(C_)
    orc.hold_update(st, n, h, gate=[1, 0, 1])
    assert h["count"].tolist() == [300, 0, 300]
    assert h["level_sum"][0] == 300 * 255 and (int(h["level_sum"][0]) & 0xFFFF) == (300 * 255) % 65536
    assert h["level_max"].tolist() == [255, 0, 255] and h["level_min"].tolist() == [255, 255, 255]
    assert h["peak_hold"][0] == 0 and h["peak_hold"][2] == orc.decode_table(8)[0xFF]


def test_window_restatement_against_plain_python(orc):
    """orc_window_update (the C restatement of the ED-137 gated window: keeplogAudioLevel Functions.cpp:2126-2145 under the
    PTT / SQU bits of Functions.cpp:1136 / 1160, and the consecutive-silence run of TransportAdapter.cpp:657-673) against a
    second, plain-Python statement of the same rules — every gate mode, closed channel gates, EMPTY records, short frames that
    leave the run alone, an alarm length small enough to fire.  The reference holds no vector for this; the ED-137 masks are pinned
    to its source text."""
    rng = np.random.default_rng(1)
    F_, C_, n = 40, 7, 160
    pl = orc.gen_uniform(F_ * C_ * n).reshape(F_, C_, n).copy()
    pl[rng.random((F_, C_)) < 0.5] = 0xD5
    st = orc.decode_meter(pl, np.full(C_, 8, np.uint8))
    info = np.zeros((F_, C_), orc.RTP_INFO)
    info["ed137"] = rng.integers(0, 2 ** 32, size=(F_, C_), dtype=np.uint64).astype(np.uint32)
    info["payload_len"] = np.where(rng.random((F_, C_)) < 0.2, 40, 160)
    st["flags"][rng.random((F_, C_)) < 0.1] |= orc.FLAG_EMPTY
    gate = (np.arange(C_) % 3 != 0).astype(np.uint8)
    for mode in range(4):
        hold, probe = orc.hold_new(C_), np.zeros(C_, orc.CHAN_PROBE)
        orc.window_update(st, hold, info=info, gate_mode=mode, alarm=3, gate=gate, probe=probe)
        for c in range(C_):
            run = al = cnt = lsum = smp = ss = pk = mx = ns = nc = 0
            mn = 255
            for f in range(F_):
                s = st[f, c]
                if s["flags"] & orc.FLAG_EMPTY:
                    continue
                l = min(int(info["payload_len"][f, c]), n)
                if l > 48:
                    if s["flags"] & orc.FLAG_PROBE_D5:
                        run += 1
                        al += run == 3
                    else:
                        run = 0
                ed = int(info["ed137"][f, c])
                squ, ptt = (ed & 0x10000000) >> 28, (ed & 0xe0000000) >> 29           # Functions.cpp:1160, 1136
                g = {0: 1, 1: squ, 2: int(ptt != 0), 3: int(squ or ptt != 0)}[mode]
                if not gate[c] or not g:
                    continue
                cnt += 1; lsum += int(s["byte_mean"]); smp += l; ss += int(s["sumsq"]); pk = max(pk, int(s["peak"]))
                mx = max(mx, int(s["byte_mean"])); mn = min(mn, int(s["byte_mean"]))
                ns += bool(s["flags"] & orc.FLAG_SILENT); nc += bool(s["flags"] & orc.FLAG_CLIPPED)
            h = hold[c]
            got = tuple(int(h[k]) for k in ("count", "level_sum", "samples", "sumsq_acc", "peak_hold", "level_max", "level_min", "n_silent", "n_clipped"))
            assert got == (cnt, lsum, smp, ss, pk, mx, mn, ns, nc), (mode, c)
            assert (int(probe["run"][c]), int(probe["alarms"][c])) == (run, al), (mode, c)
        assert probe["alarms"].sum() > 0
