"""TEST INFRASTRUCTURE ONLY — Python handle on the CPU oracle.

Loads ``oracle/_build/libigdsp_oracle.so`` (built by ``oracle/Makefile`` from
``igdsp_oracle.c``) and exposes numpy-typed wrappers.  Also holds an independent
numpy restatement of the G.711 expansion (``np_ulaw2lin`` / ``np_alaw2lin``) so
the C oracle is cross-checked by a second implementation besides the audioop
fixtures.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  See igdsp_oracle.h for the parity status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libigdsp_oracle.so")
_REF_WAV_PATH = os.path.join(_HERE, "_ref", "libref_wavwriter.so")
_REF_METER_PATH = os.path.join(_HERE, "_ref", "libref_audiometer.so")

FRAME_STATS = np.dtype(
    [("sumsq", "<u8"), ("rms", "<f4"), ("peak", "<u2"), ("byte_mean", "u1"), ("flags", "u1")], align=True
)
CHAN_HOLD = np.dtype(
    [
        ("sumsq_acc", "<u8"), ("count", "<u4"), ("level_sum", "<u4"), ("samples", "<u4"),
        ("peak_hold", "<u2"), ("level_max", "u1"), ("level_min", "u1"),
        ("n_silent", "<u4"), ("n_clipped", "<u4"),
    ],
    align=True,
)
AGGREGATE = np.dtype({
    "names": ["sumsq", "samples", "frames", "n_silent", "n_clipped", "byte_mean_sum", "peak_slot"],
    "formats": ["<u8", "<u8", "<u8", "<u8", "<u8", "<u8", ("<u8", (8,))],
    "offsets": [128 * i for i in range(7)],
    "itemsize": 896,
})
assert FRAME_STATS.itemsize == 16 and CHAN_HOLD.itemsize == 32 and AGGREGATE.itemsize == 896

RTP_INFO = np.dtype([("ed137", "<u4"), ("payload_len", "<u2"), ("pt", "u1"), ("flags", "u1")], align=True)
assert RTP_INFO.itemsize == 8
CHAN_PROBE = np.dtype([("run", "<u4"), ("alarms", "<u4")], align=True)
GATE_ALWAYS, GATE_SQU, GATE_PTT, GATE_SQU_OR_PTT = 0, 1, 2, 3
FLAG_SILENT, FLAG_PROBE_D5, FLAG_CLIPPED, FLAG_EMPTY = 1, 2, 4, 8
ENC_SUN16, ENC_G191 = 0, 1
SEED = 0x20241218


def build(force: bool = False) -> str:
    """Compile the C oracle (and oracle/_ref when /root/reference is mounted)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "igdsp_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-s", "_build/libigdsp_oracle.so"], check=True)
    def _ref_stale():
        outs = [_REF_WAV_PATH, _REF_METER_PATH]
        if not all(os.path.exists(o) for o in outs):
            return True
        newest_src = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ref_wavwriter_shim.cpp", "ref_audiometer_shim.cpp", "Makefile"))
        return min(os.path.getmtime(o) for o in outs) < newest_src

    if os.path.isdir("/root/reference") and (force or _ref_stale()):
        subprocess.run(["make", "-C", _HERE, "-s", "ref"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p, u16p, i16p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_int16), C.c_void_p
        L.orc_ulaw2lin.restype = C.c_int16; L.orc_ulaw2lin.argtypes = [C.c_uint8]
        L.orc_alaw2lin.restype = C.c_int16; L.orc_alaw2lin.argtypes = [C.c_uint8]
        L.orc_lin2ulaw.restype = C.c_uint8; L.orc_lin2ulaw.argtypes = [C.c_int16, C.c_int]
        L.orc_lin2alaw.restype = C.c_uint8; L.orc_lin2alaw.argtypes = [C.c_int16, C.c_int]
        L.orc_byte_mean.restype = C.c_uint8; L.orc_byte_mean.argtypes = [vp, C.c_int]
        L.orc_byte_mean_signed_char.restype = C.c_uint8; L.orc_byte_mean_signed_char.argtypes = [vp, C.c_int]
        L.orc_percent.restype = C.c_int; L.orc_percent.argtypes = [C.c_double]
        L.orc_decode_meter.restype = None
        L.orc_decode_meter.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint32]
        L.orc_encode.restype = None
        L.orc_encode.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_int]
        L.orc_hold_reset.restype = None; L.orc_hold_reset.argtypes = [vp, C.c_uint32, vp]
        L.orc_hold_update.restype = None
        L.orc_hold_update.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
        L.orc_window_update.restype = None
        L.orc_window_update.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
        L.orc_roundtrip_peakhold.restype = None
        L.orc_roundtrip_peakhold.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_int]
        L.orc_depayload.restype = None
        L.orc_depayload.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
        L.orc_g726_reorder.restype = None; L.orc_g726_reorder.argtypes = [vp, vp, C.c_size_t, C.c_int]
        L.orc_splitmix64.restype = C.c_uint64; L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_gen_uniform.restype = None; L.orc_gen_uniform.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_gen_speech.restype = None
        L.orc_gen_speech.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int]
        L.orc_wav_header.restype = C.c_size_t; L.orc_wav_header.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.orc_wav_expand.restype = None; L.orc_wav_expand.argtypes = [vp, C.c_uint32, vp]
        L.orc_time_decode_meter.restype = C.c_double
        L.orc_time_decode_meter.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp]
        L.orc_time_single_frame.restype = C.c_double
        L.orc_time_single_frame.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp]
        L.orc_time_byte_mean.restype = C.c_double
        L.orc_time_byte_mean.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- G.711 scalar / table
def decode_table(pt: int) -> np.ndarray:
    L = lib()
    f = L.orc_alaw2lin if pt == 8 else L.orc_ulaw2lin
    return np.array([f(i) for i in range(256)], dtype="<i2")


def encode_table(pt: int, variant: int) -> np.ndarray:
    """code for every int16 input, indexed by (pcm + 32768)."""
    C_, F_, n = 1, 256, 256
    pcm = np.arange(-32768, 32768, dtype="<i2").reshape(F_, C_, n)
    return encode(pcm, np.array([pt], dtype=np.uint8), variant).reshape(-1)


def np_ulaw2lin(codes: np.ndarray) -> np.ndarray:
    """Independent numpy restatement of ITU-T G.711 mu-law expansion."""
    u = (~codes.astype(np.uint8)).astype(np.int32)
    t = (((u & 0x0F) << 3) + 0x84) << ((u & 0x70) >> 4)
    return np.where(u & 0x80, 0x84 - t, t - 0x84).astype(np.int16)


def np_alaw2lin(codes: np.ndarray) -> np.ndarray:
    """Independent numpy restatement of ITU-T G.711 A-law expansion."""
    a = codes.astype(np.int32) ^ 0x55
    seg = (a & 0x70) >> 4
    t = (a & 0x0F) << 4
    t = np.where(seg == 0, t + 8, (t + 0x108) << np.maximum(seg - 1, 0))
    return np.where(a & 0x80, t, -t).astype(np.int16)


# --------------------------------------------------------------------------- batched
def decode_meter(payload, codec, length=None, want_pcm=False, want_agg=False, rank=0):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    F_, C_, n = payload.shape
    codec = np.ascontiguousarray(codec, dtype=np.uint8)
    assert codec.shape == (C_,)
    stats = np.zeros((F_, C_), dtype=FRAME_STATS)
    pcm = np.zeros((F_, C_, n), dtype="<i2") if want_pcm else None
    agg = np.zeros((), dtype=AGGREGATE) if want_agg else None
    if length is not None:
        length = np.ascontiguousarray(length, dtype="<u2")
    lib().orc_decode_meter(_p(payload), _p(codec), _p(length), C_, F_, n, _p(stats), _p(pcm), _p(agg), rank)
    out = [stats]
    if want_pcm:
        out.append(pcm)
    if want_agg:
        out.append(agg)
    return out[0] if len(out) == 1 else tuple(out)


def encode(pcm, codec, variant=ENC_G191):
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    F_, C_, n = pcm.shape
    codec = np.ascontiguousarray(codec, dtype=np.uint8)
    out = np.zeros((F_, C_, n), dtype=np.uint8)
    lib().orc_encode(_p(pcm), _p(codec), C_, F_, n, _p(out), variant)
    return out


def hold_new(C_):
    h = np.zeros((C_,), dtype=CHAN_HOLD)
    h["level_min"] = 255
    return h


def hold_update(stats, n, hold, gate=None):
    F_, C_ = stats.shape
    if gate is not None:
        gate = np.ascontiguousarray(gate, dtype=np.uint8)
    lib().orc_hold_update(_p(np.ascontiguousarray(stats)), C_, F_, n, _p(hold), _p(gate))
    return hold


def window_update(stats, hold, info=None, length=None, n=160, gate_mode=GATE_ALWAYS, alarm=0, gate=None, probe=None):
    """Fold stats[F][C] into hold[C] under per-frame ED-137 gates, and follow the consecutive-silence run in probe[C]
    (orc_window_update, oracle/igdsp_oracle.h).  hold / probe are updated in place and returned."""
    stats = np.ascontiguousarray(stats)
    F_, C_ = stats.shape
    if info is not None:
        info = np.ascontiguousarray(info, dtype=RTP_INFO)
    if length is not None:
        length = np.ascontiguousarray(length, dtype="<u2")
    if gate is not None:
        gate = np.ascontiguousarray(gate, dtype=np.uint8)
    lib().orc_window_update(_p(stats), _p(info), _p(length), C_, F_, n, gate_mode, alarm, _p(hold), _p(gate), _p(probe))
    return hold, probe


def roundtrip_peakhold(payload, codec, hold, gate=None, variant=ENC_G191):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    F_, C_, n = payload.shape
    codec = np.ascontiguousarray(codec, dtype=np.uint8)
    out = np.zeros_like(payload)
    stats = np.zeros((F_, C_), dtype=FRAME_STATS)
    if gate is not None:
        gate = np.ascontiguousarray(gate, dtype=np.uint8)
    lib().orc_roundtrip_peakhold(_p(payload), _p(codec), C_, F_, n, _p(out), _p(stats), _p(hold), _p(gate), variant)
    return out, stats, hold


def depayload(packets, sizes, radio, n=160):
    """packets [F][C][stride] u8 -> (payload [F][C][n], len [F][C] u16, info [F][C])."""
    packets = np.ascontiguousarray(packets, dtype=np.uint8)
    F_, C_, stride = packets.shape
    radio = np.ascontiguousarray(radio, dtype=np.uint8)
    if sizes is not None:
        sizes = np.ascontiguousarray(sizes, dtype="<u2")
    payload = np.zeros((F_, C_, n), np.uint8)
    ln = np.zeros((F_, C_), "<u2")
    info = np.zeros((F_, C_), RTP_INFO)
    lib().orc_depayload(_p(packets), _p(sizes), _p(radio), C_, F_, stride, n, _p(payload), _p(ln), _p(info))
    return payload, ln, info


def g726_reorder(data: np.ndarray, mode: int) -> np.ndarray:
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    out = np.zeros_like(data)
    lib().orc_g726_reorder(_p(data), _p(out), data.size, mode)
    return out


def byte_mean(buf: bytes | np.ndarray, signed_char=False) -> int:
    a = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, np.uint8)
    f = lib().orc_byte_mean_signed_char if signed_char else lib().orc_byte_mean
    return int(f(_p(a), a.size))


def percent(level: float) -> int:
    return int(lib().orc_percent(float(level)))


# --------------------------------------------------------------------------- synthetic data
def gen_uniform(n_bytes: int, seed: int = SEED, first_byte: int = 0) -> np.ndarray:
    out = np.empty((n_bytes,), dtype=np.uint8)
    lib().orc_gen_uniform(_p(out), n_bytes, seed, first_byte)
    return out


def gen_speech(C_, F_, n, codec, seed: int = SEED, first_channel: int = 0, variant=ENC_G191) -> np.ndarray:
    codec = np.ascontiguousarray(codec, dtype=np.uint8)
    out = np.empty((F_, C_, n), dtype=np.uint8)
    lib().orc_gen_speech(_p(out), _p(codec), C_, F_, n, seed, first_channel, variant)
    return out


def wav_header(rate: int, data_bytes: int) -> bytes:
    out = np.zeros((44,), dtype=np.uint8)
    lib().orc_wav_header(_p(out), rate, data_bytes)
    return out.tobytes()


def wav_expand(payload: np.ndarray) -> np.ndarray:
    payload = np.ascontiguousarray(payload, dtype=np.uint8).reshape(-1)
    out = np.zeros((payload.size * 2,), dtype=np.uint8)
    lib().orc_wav_expand(_p(payload), payload.size, _p(out))
    return out


# --------------------------------------------------------------------------- cpu_baseline timing
def time_decode_meter(payload, codec, threads: int, reps: int = 1) -> float:
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    F_, C_, n = payload.shape
    stats = np.zeros((F_, C_), dtype=FRAME_STATS)
    return float(lib().orc_time_decode_meter(_p(payload), _p(np.ascontiguousarray(codec, np.uint8)), C_, F_, n, threads, reps, _p(stats)))


def time_single_frame(payload, codec, threads: int, reps: int = 1, return_slots: bool = False):
    """B1 as BASELINE.md section 2 words it: one frame per call through the single-frame shim; seconds for all reps."""
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    F_, C_, n = payload.shape
    slots = np.zeros((C_,), dtype=FRAME_STATS)
    t = float(lib().orc_time_single_frame(_p(payload), _p(np.ascontiguousarray(codec, np.uint8)), C_, F_, n, threads, reps, _p(slots)))
    return (t, slots) if return_slots else t


def time_byte_mean(payload, threads: int, reps: int = 1) -> float:
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    F_, C_, n = payload.shape
    out = np.zeros((F_, C_), dtype=np.uint8)
    return float(lib().orc_time_byte_mean(_p(payload), C_, F_, n, threads, reps, _p(out)))


# --------------------------------------------------------------------------- real reference recorder
def ref_wavwriter_available() -> bool:
    return os.path.exists(_REF_WAV_PATH)


def ref_wav_record(tmpdir: str, payloads: np.ndarray, rate: int = 8000) -> bytes:
    """Run the REAL reference WavWriter (oracle/_ref) over payload frames; return file bytes."""
    L = C.CDLL(_REF_WAV_PATH)
    payloads = np.ascontiguousarray(payloads, dtype=np.uint8)
    nf, plen = payloads.shape
    pkt = np.zeros((12,), dtype=np.uint8)
    pkt[0] = 0x80
    import tempfile

    # names are <prefix>YYYYMMDDhhmmss.wav (WavWriter.cpp:197-220): one fresh dir per recording
    sub = tempfile.mkdtemp(dir=tmpdir)
    prefix = os.path.join(sub, "")
    rc = L.ref_wav_record(prefix.encode(), b"ref", rate, _p(pkt), 12, _p(payloads), nf, plen)
    assert rc == 0
    new = sorted(os.listdir(sub))
    assert len(new) == 1, new
    with open(os.path.join(sub, new[0]), "rb") as fh:
        return fh.read()


def ref_audiometer_available() -> bool:
    return os.path.exists(_REF_METER_PATH)


def ref_audiometer_percent(levels, card: str = "igdsp") -> list:
    """Feed integer levels through the REAL reference AudioMeter::getAudioLevel() (oracle/_ref, built from
    /root/reference/audiometer.cpp + its moc output) via its FIFO; returns the percents it emitted."""
    L = C.CDLL(_REF_METER_PATH)
    L.ref_audiometer_percent.restype = C.c_int
    lv = np.ascontiguousarray(levels, dtype=np.int32)
    out = np.zeros_like(lv)
    n = L.ref_audiometer_percent(f"{card}{os.getpid()}".encode(), _p(lv), lv.size, _p(out))
    assert n == lv.size, (n, lv.size)
    return out.tolist()
