#!/usr/bin/env python3
"""igdsp_io_alloc, many fresh processes, several buffer sets: does every run find its classes?  (GPU box)
usage: io_soak.py [runs]      one line per run: shape, report, probe time input -> each output half"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import torch
from igate4xsoftphonedsp_amd import capi
in_b, rec_b, bulk_b = (int(x) for x in sys.argv[1:4])
torch.cuda.set_device(0)
ctx = capi.Context(0, 1024)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); hs = s.cuda_stream
bufs = [(in_b, capi.IO_INPUT), (rec_b, capi.IO_RECORD)] + ([(bulk_b, capi.IO_BULK)] if bulk_b else [])
st, ptrs, rep = ctx.io_alloc(bufs)
n = min(in_b, 1342177280) // 10240 * 10240
out = {"rep": {k: rep[k] for k in ("placed", "bulk_spread", "classes_found", "chunks_explored", "probes", "reseeds")}, "setup_ms": round(rep["setup_ms"]), "settle_ms": round(rep["settle_ms"])}
for _ in range(3): ctx.probe_placement(ptrs[0], n, out=ptrs[1], reps=10, stream=hs)
out["in->rec"] = round(min(ctx.probe_placement(ptrs[0], n, out=ptrs[1], reps=10, stream=hs) for _ in range(3)), 4)
if bulk_b:
    half = (bulk_b // 2) // (128 << 20) * (128 << 20)
    out["in->bulk0"] = round(min(ctx.probe_placement(ptrs[0], n, out=ptrs[2], reps=10, stream=hs) for _ in range(3)), 4)
    out["in->bulk1"] = round(min(ctx.probe_placement(ptrs[0], n, out=ptrs[2] + bulk_b - (160 << 20), reps=10, stream=hs) for _ in range(3)), 4)
    out["bulk0->bulk1"] = round(min(ctx.probe_placement(ptrs[2], min(n, half // 10240 * 10240), out=ptrs[2] + bulk_b - (160 << 20), reps=10, stream=hs) for _ in range(3)), 4)
print(json.dumps(out))
''' % ROOT
B = 65536 * 128
SHAPES = {"store160": (B * 160, B * 16, B * 320), "store164": (B * 164, B * 16, B * 328), "roundtrip": (B * 160, B * 16, B * 160),
          "encode": (B * 320, B * 16, B * 160), "store80": (B * 80, B * 16, B * 160), "store240": (B * 240, B * 16, B * 480)}
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2:
    SHAPES = {k: v for k, v in SHAPES.items() if k in sys.argv[2:]}
os.makedirs(os.path.join(ROOT, "gpurun_out", "io_soak"), exist_ok=True)
for r in range(runs):
    for name, (a, b, c) in SHAPES.items():
        env = dict(os.environ, IGDSP_IO_DEBUG="1")
        p = subprocess.run([sys.executable, "-c", CHILD, str(a), str(b), str(c)], capture_output=True, text=True, env=env, timeout=120)
        log = os.path.join(ROOT, "gpurun_out", "io_soak", f"{name}_{r}.log")
        open(log, "w").write(p.stderr)
        line = p.stdout.strip().split("\n")[-1] if p.stdout.strip() else f"FAILED rc={p.returncode}"
        print(name, r, line, flush=True)
