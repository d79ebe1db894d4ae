#!/usr/bin/env python3
"""igdsp_io_alloc on the GPU box: the report, and what the placed buffers buy the real kernels against plain consecutive
allocations in the SAME process.  usage: io_place.py [meter|store|roundtrip|encode ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igate4xsoftphonedsp_amd import capi

C_, F_, n = 65536, 128, 160
modes = sys.argv[1:] or ["meter", "store"]
torch.cuda.set_device(0)
ctx = capi.Context(0, 1024)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); hs = s.cuda_stream
B = F_ * C_ * n

def gpu_ms(fn, reps):
    t = ctx.timer(); t.start(hs)
    for _ in range(reps): fn()
    t.stop(hs); ms = t.elapsed_ms() / reps; t.close(); return ms

def run(mode, pl, st, bulk, cd, hold):
    if mode == "meter": return lambda: ctx.decode_meter(pl, cd, C_, F_, n, st, stream=hs)
    if mode == "store": return lambda: ctx.decode_meter(pl, cd, C_, F_, n, st, pcm=bulk, stream=hs)
    if mode == "roundtrip": return lambda: ctx.roundtrip_peakhold(pl, cd, C_, F_, n, bulk, st, hold, stream=hs)
    if mode == "encode": return lambda: ctx.encode(pl, cd, C_, F_, n, bulk, stream=hs)

import ctypes as CT
for mode in (0, 1):
    b3 = (CT.c_int * 3)(-1, -1, -1)
    rc = ctx.L.igdsp_internal_vmm_remap_check(ctx.h, mode, b3)
    print("vmm_remap_check mode", mode, "rc", rc, "X byte", hex(b3[0]), "(expect 0x11)  Y byte", hex(b3[1]), "(expect 0x22)  same address", b3[2], flush=True)
res = {}
for mode in modes:
    in_b = B * (2 if mode == "encode" else 1)
    bulk_b = {"meter": 0, "store": 2 * B, "roundtrip": B, "encode": B}[mode]
    cd = torch.zeros((C_,), dtype=torch.uint8, device="cuda")
    hold = torch.zeros((C_ * 32,), dtype=torch.uint8, device="cuda")
    ctx.hold_reset(hold, C_, stream=hs)
    # naive: plain consecutive allocations through the ABI
    p_in = ctx.dev_alloc(in_b); p_st = ctx.dev_alloc(F_ * C_ * 16); p_bulk = ctx.dev_alloc(bulk_b) if bulk_b else None
    ctx.gen_uniform(p_in, in_b, stream=hs)
    f = run(mode, p_in, p_st, p_bulk, cd, hold)
    for _ in range(3): gpu_ms(f, 50)
    t_naive = min(gpu_ms(f, 50) for _ in range(3))
    # placed
    t0 = time.time()
    bufs = [(in_b, capi.IO_INPUT), (F_ * C_ * 16, capi.IO_RECORD)] + ([(bulk_b, capi.IO_BULK)] if bulk_b else [])
    st_, ptrs, rep = ctx.io_alloc(bufs)
    rep["wall_s"] = round(time.time() - t0, 2)
    pz = min(ctx.probe_placement(ptrs[0], B, out=ptrs[1], reps=10, stream=hs) for _ in range(3))
    print("probe placed, input as allocated (zeros): %.4f ms" % pz, flush=True)
    ctx.gen_uniform(ptrs[0], in_b, stream=hs)
    pz = min(ctx.probe_placement(ptrs[0], B, out=ptrs[1], reps=10, stream=hs) for _ in range(3))
    print("probe placed, input random: %.4f ms" % pz, flush=True)
    g = run(mode, ptrs[0], ptrs[1], ptrs[2] if bulk_b else None, cd, hold)
    for _ in range(3): gpu_ms(g, 50)
    t_placed = min(gpu_ms(g, 50) for _ in range(3))
    t_naive2 = min(gpu_ms(f, 50) for _ in range(3))
    for reps in (2, 4, 10, 50):
        a = min(ctx.probe_placement(ptrs[0], B, out=ptrs[1], reps=reps, stream=hs) for _ in range(3))
        b = min(ctx.probe_placement(ptrs[0], B, out=ptrs[1], reps=reps, stream=None) for _ in range(3))
        c = min(gpu_ms(g, reps) for _ in range(3))
        print("reps %d: probe on torch stream %.4f  on null stream %.4f   kernel %.4f ms" % (reps, a, b, c), flush=True)
    pr_placed = min(ctx.probe_placement(ptrs[0], B, out=ptrs[1], reps=10, stream=hs) for _ in range(3))
    pr_naive = min(ctx.probe_placement(p_in, B, out=p_st, reps=10, stream=hs) for _ in range(3))
    print("probe (whole batch -> records): placed %.4f ms  naive %.4f ms   va in %#x st %#x" % (pr_placed, pr_naive, ptrs[0], ptrs[1]), flush=True)
    # read-only stream over both inputs (is VMM-mapped memory as fast to read as hipMalloc memory?)
    sink = torch.zeros((1,), dtype=torch.int64, device="cuda")
    r_naive = gpu_ms(lambda: ctx.stream_read(p_in, in_b, sink, stream=hs), 20)
    r_placed = gpu_ms(lambda: ctx.stream_read(ptrs[0], in_b, sink, stream=hs), 20)
    res[mode] = {"ms_naive": round(t_naive, 4), "ms_naive_again": round(t_naive2, 4), "ms_placed": round(t_placed, 4),
                 "gain": round(t_naive / t_placed, 4), "read_GBs_naive": round(in_b / r_naive / 1e6, 1), "read_GBs_placed": round(in_b / r_placed / 1e6, 1),
                 "report": rep}
    print(mode, json.dumps(res[mode]), flush=True)
    st_.close()
    ctx.dev_free(p_in); ctx.dev_free(p_st)
    if p_bulk: ctx.dev_free(p_bulk)
print(json.dumps(res))
ctx.close()
