"""ctypes view of the C++ host mirror (libigdsp_host.so): tp_adapter / trx layouts of igdsp_host.h."""
import ctypes as C
import os

from igate4xsoftphonedsp_amd import build as igbuild
from igate4xsoftphonedsp_amd import capi


class TpAdapter(C.Structure):
    _fields_ = [
        ("stream_user_data", C.c_void_p), ("stream_rtp_cb", C.c_void_p), ("radiostatus", C.c_int), ("rtpFalse", C.c_int16),
        ("callID", C.c_int), ("ed137_value", C.c_uint32), ("payloadsize", C.c_uint32), ("pkt_buff", C.c_uint8 * 256),
        ("bufSize", C.c_size_t), ("payload_buff", C.c_uint8 * 256), ("payload_bufSize", C.c_size_t),
        ("send_pkt_buff", C.c_uint8 * 256), ("tmp_payload_buf", C.c_uint8 * 256), ("send_bufSize", C.c_size_t),
        ("send_payload_buff", C.c_uint8 * 256), ("send_payload_bufSize", C.c_size_t), ("r2sPacket", C.c_longlong),
        ("rtpAudio", C.c_int), ("last_rx_pt", C.c_uint8), ("last_tx_pt", C.c_uint8),
    ]


class Trx(C.Structure):
    _fields_ = [
        ("call_id", C.c_int), ("OutgoingRTP", C.c_uint8), ("IncomingRTP", C.c_uint8), ("in_rms", C.c_float), ("out_rms", C.c_float),
        ("in_peak", C.c_uint16), ("out_peak", C.c_uint16), ("in_peak_hold", C.c_uint16), ("out_peak_hold", C.c_uint16),
        ("in_percent", C.c_int), ("out_percent", C.c_int), ("in_flags", C.c_uint8), ("out_flags", C.c_uint8),
    ]


class PttWindow(C.Structure):
    _fields_ = [("eventPttSQL_In_LoggingOn", C.c_bool), ("level_in_count", C.c_int), ("level_in", C.c_double),
                ("level_in_av", C.c_double), ("level_in_max", C.c_double), ("level_in_min", C.c_double),
                ("OutgoingRTPSum", C.c_uint16), ("OutgoingRTPav", C.c_uint8), ("OutgoingRTPmax", C.c_uint8),
                ("OutgoingRTPmin", C.c_uint8)]


STREAM_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_long)
_lib = None


def load():
    global _lib
    if _lib is None:
        igbuild.build()
        capi.load()                       # torch's HIP runtime first, then libigdsp.so (see capi.load)
        L = C.CDLL(igbuild.HOST_LIB)
        L.igdsp_host_create.restype = C.c_void_p; L.igdsp_host_create.argtypes = [C.c_int, C.c_uint32]
        L.igdsp_host_destroy.restype = None; L.igdsp_host_destroy.argtypes = [C.c_void_p]
        L.igdsp_host_adapter_new.restype = C.POINTER(TpAdapter); L.igdsp_host_adapter_new.argtypes = [C.c_int, C.c_int]
        L.igdsp_host_adapter_free.restype = None; L.igdsp_host_adapter_free.argtypes = [C.POINTER(TpAdapter)]
        L.igdsp_host_bind_radio.restype = C.c_int; L.igdsp_host_bind_radio.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.igdsp_host_set_mode.restype = C.c_int; L.igdsp_host_set_mode.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.igdsp_host_tick.restype = C.c_int; L.igdsp_host_tick.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.igdsp_host_get_trx.restype = C.c_int; L.igdsp_host_get_trx.argtypes = [C.c_void_p, C.c_int, C.POINTER(Trx)]
        L.igdsp_host_ed137_events.restype = C.c_uint32; L.igdsp_host_ed137_events.argtypes = [C.c_void_p]
        L.igdsp_host_keeplog.restype = C.c_int; L.igdsp_host_keeplog.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.igdsp_host_ptt_event.restype = C.c_int
        L.igdsp_host_ptt_event.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_double, C.c_char_p, C.c_size_t]
        L.igdsp_host_get_window.restype = C.c_int; L.igdsp_host_get_window.argtypes = [C.c_void_p, C.c_int, C.POINTER(PttWindow)]
        L.transport_rtp_cb.restype = None; L.transport_rtp_cb.argtypes = [C.POINTER(TpAdapter), C.c_void_p, C.c_long]
        L.transport_send_rtp.restype = C.c_int; L.transport_send_rtp.argtypes = [C.POINTER(TpAdapter), C.c_void_p, C.c_size_t]
        L.igdsp_meter_fifo_open.restype = C.c_int; L.igdsp_meter_fifo_open.argtypes = [C.c_char_p, C.c_int]
        L.igdsp_meter_fifo_write.restype = C.c_int; L.igdsp_meter_fifo_write.argtypes = [C.c_int, C.c_int]
        L.igdsp_meter_fifo_close.restype = C.c_int; L.igdsp_meter_fifo_close.argtypes = [C.c_int]
        L.igdsp_wav_start.restype = C.c_void_p; L.igdsp_wav_start.argtypes = [C.c_char_p, C.c_int]
        L.igdsp_wav_writeRTPWav.restype = C.c_int
        L.igdsp_wav_writeRTPWav.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
        L.igdsp_wav_stop.restype = C.c_int; L.igdsp_wav_stop.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def rtp_packet(pt: int, seq: int, payload: bytes, radio: bool, ed137_word: int = 0) -> bytes:
    """12-byte RTP header (V=2), plus the 8-byte ED-137 extension (profile 0x0167, len 1, word) on radio calls."""
    import struct

    b0 = 0x80 | (0x10 if radio else 0)
    hdr = struct.pack("!BBHII", b0, pt & 0x7F, seq & 0xFFFF, (seq * 160) & 0xFFFFFFFF, 0x1234ABCD)
    if radio:
        hdr += struct.pack("!HHI", 0x0167, 1, ed137_word & 0xFFFFFFFF)
    return hdr + payload
