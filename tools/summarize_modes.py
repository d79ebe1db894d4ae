#!/usr/bin/env python3
"""gpurun_out/prof_modes/<mode>/**/kernel_trace.csv -> profiles/<tag>_modes_kernel_trace.md: for each bench mode, the
dominant igdsp kernel's durations over ALL dispatches and over the timed region (the last 20 dispatches)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
KEY = {"store": "k_meter_chunk64<true", "roundtrip": "k_roundtrip_chunk64", "rtp": "k_meter_rtp64", "packets": "k_meter_rtp64",
       "depayload": "k_depayload64", "encode": "k_encode_lut16"}
lines = [f"# rocprofv3 --kernel-trace of the secondary bench modes — {tag} (`tools/profile_modes.sh`: `bench.py --mode M --steps 20 --warmup 3`)", "",
         "| mode | kernel | dispatches | avg over all (pre-warm, placement trials, warm-up, timed) | timed region: last 20, avg | min | max |", "|---|---|---|---|---|---|---|"]
for mode, key in KEY.items():
    g = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_modes", mode, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not g:
        continue
    d = []
    for r in csv.DictReader(open(g[-1])):
        if key in r["Kernel_Name"]:
            d.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    d = [x[1] for x in sorted(d)]
    if not d:
        continue
    t = d[-20:]
    lines.append(f"| {mode} | `{key}` | {len(d)} | {sum(d) / len(d) / 1e3:.1f} us | **{sum(t) / len(t) / 1e3:.1f} us** | {min(t) / 1e3:.1f} | {max(t) / 1e3:.1f} |")
open(os.path.join(ROOT, "profiles", f"{tag}_modes_kernel_trace.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
