#!/usr/bin/env python3
"""Round-3 A/B: does clustering the record stores shrink the same-class read / write penalty?  (GPU box)
The meter's traffic (10 KiB read + 1 KiB record block per super-chunk of 64 frames, no per-sample work) with the record blocks of
K consecutive super-chunks stored as one K KiB run per wave (k_stream_cluster<K>; K = 1 is k_stream_rw), writing into
  same : a buffer of the INPUT's memory class (what consecutive plain allocations normally give),
  other: a buffer of another class (what igdsp_io_alloc gives),
plus the product kernel on both.  Alternating passes in one process; median of 8 groups of 40 launches."""
import ctypes as CT, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igate4xsoftphonedsp_amd import capi

C_, F_ = 65536, 128
B = C_ * F_
torch.cuda.set_device(0)
ctx = capi.Context(0, 1024)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); hs = s.cuda_stream
in_b = B * 160
# two INPUT-role buffers share the inputs' class; the RECORD-role buffer sits in another class
st, ptrs, rep = ctx.io_alloc([(in_b, capi.IO_INPUT), (B * 16, capi.IO_INPUT), (B * 16, capi.IO_RECORD)])
print(json.dumps({k: rep[k] for k in ("placed", "classes_found", "probe_ms_same", "probe_ms_other", "settle_ms")}), flush=True)
ctx.gen_uniform(ptrs[0], in_b, stream=hs)
cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
agg = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")
fc = ctx.L.igdsp_internal_stream_cluster
fc.restype = CT.c_int
fc.argtypes = [CT.c_void_p, CT.c_void_p, CT.c_size_t, CT.c_void_p, CT.c_int, CT.c_void_p]
NB = B * 177                                    # algorithmic bytes of the meter launch


def run(f, groups=8, per=40):
    for _ in range(20):
        f()
    out = []
    for _ in range(groups):
        t = ctx.timer(); t.start(hs)
        for _ in range(per):
            f()
        t.stop(hs); out.append(t.elapsed_ms() / per); t.close()
    return statistics.median(out)


res = {}
for rnd in range(2):
    for K in (1, 2, 4, 8, 16):
        for name, dst in (("same", ptrs[1]), ("other", ptrs[2])):
            ms = run(lambda K=K, dst=dst: fc(ctx.h, ptrs[0], in_b, dst, K, hs))
            res.setdefault(f"bare K={K:2d} {name}", []).append(ms)
    for name, dst in (("same", ptrs[1]), ("other", ptrs[2])):
        ms = run(lambda dst=dst: ctx.decode_meter(ptrs[0], cd, C_, F_, 160, dst, agg=agg, rank=0, stream=hs))
        res.setdefault(f"k_meter_chunk64 {name}", []).append(ms)
for k, v in res.items():
    m = min(v)
    print("%-26s %s ms  best %.4f ms = %.3f of 8 TB/s" % (k, " ".join("%.4f" % x for x in v), m, NB / (m * 1e-3) / 8e12), flush=True)
print(json.dumps({k: [round(x, 4) for x in v] for k, v in res.items()}))
