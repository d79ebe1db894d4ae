#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into the tracked profiles/<tag>_* files:
  <tag>_kernel_stats.csv      rocprofv3 --stats table of the headline run (ALL dispatches, warm-up and placement probes included)
  <tag>_kernel_timed.csv      per mode: the dominant kernel over the TIMED REGION only (the last K dispatches, K = --steps)
  <tag>_pmc_<mode>_<pass>.csv raw counter files (fetch / write / sq / sq2)
  <tag>_summary.md            everything above in words, plus the un-profiled bench lines
  pmc_traffic.json            {mode: HBM bytes per launch} read by bench.py for roofline.traffic (labelled with its source)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
STEPS = 20                    # tools/profile.sh runs bench.py --steps 20: the LAST 20 dispatches of the kernel are the timed region
C_, F_ = 65536, 128
KERNEL = {"meter": "k_meter_chunk64", "store": "k_meter_chunk64", "roundtrip": "k_roundtrip_blk64", "depayload": "k_depayload64",
          "rtp": "k_meter_rtp64", "packets": "k_meter_rtp64", "window": "k_meter_rtp64", "encode": "k_encode_lut16", "wav": "k_wav_expand16", "meter164": "k_meter_strided",
          "store164": "k_meter_strided", "roundtrip164": "k_roundtrip_strided", "meter24": "k_meter_tiny"}
BPS = {"meter": (160 + 1 + 16) / 160, "store": (160 + 1 + 16 + 320) / 160, "roundtrip": (160 + 1 + 16 + 160) / 160,
       "depayload": (180 + 160 + 2 + 8) / 160, "rtp": (192 + 1 + 16 + 8) / 160, "packets": (180 + 1 + 16 + 8) / 160, "window": (180 + 1 + 16 + 8) / 160,
       "encode": (320 + 1 + 160) / 160, "wav": 480 / 160, "meter164": (164 + 1 + 16) / 164,
       "store164": (164 + 1 + 16 + 328) / 164, "roundtrip164": (164 + 1 + 16 + 164) / 164, "meter24": (24 + 1 + 16) / 24}
SAMPLES = {m: C_ * F_ * (164 if m.endswith("164") else 24 if m.endswith("24") else 160) for m in KERNEL}


def newest(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return g[-1] if g else None


def durations(mode, sub):
    f = newest(f"{mode}/{sub}/**/*kernel_trace.csv")
    d = []
    if f:
        for r in csv.DictReader(open(f)):
            if KERNEL[mode] in r["Kernel_Name"]:
                d.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return [x[1] for x in sorted(d)]


def counters(mode, sub):
    f = newest(f"{mode}/{sub}/**/*counter_collection.csv")
    per = collections.defaultdict(list)
    if f:
        rows = [r for r in csv.DictReader(open(f)) if KERNEL[mode] in r["Kernel_Name"]]
        # timed region only: the last STEPS dispatches of the kernel
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-STEPS:]
        for r in rows:
            if int(r["Dispatch_Id"]) in ids:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        shutil.copy(f, os.path.join(dst, f"{tag}_pmc_{mode}_{sub}.csv"))
    return {k: sum(v) / len(v) for k, v in per.items()}


lines = [f"# rocprofv3 summary — {tag} (1x MI355X; every run: `bench.py --mode <m> --steps 20 --warmup 3 --placement abi` under rocprofv3)", ""]
ks = newest("meter/stats/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    lines += ["## headline run, `--kernel-trace --stats` (ALL dispatches: clock pre-warm, igdsp_io_alloc's probe stream `k_stream_rw`, warm-up, timed)", "",
              "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
    for r in list(csv.DictReader(open(ks)))[:8]:
        lines.append(f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
    lines.append("")

timed_rows, traffic = [], {}
lines += ["## per mode: dominant kernel over the TIMED REGION (last 20 dispatches), HBM traffic from separate FETCH_SIZE / WRITE_SIZE passes", "",
          "FETCH_SIZE x 2 x 1024 (gfx950 reports half the bytes of 16 B/lane streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE x 1024; the dword-aligned 16-byte loads of",
          "`packets` / `depayload` and the 8-byte LDS-staged stores are outside the guide's calibration, so their ratios are indicative.", "",
          "| mode | kernel | timed avg us (kernel trace) | min | max | algorithmic MB | PMC read MB | PMC write MB | PMC / algorithmic | bench kernel_avg_ms (un-profiled) | frac of 8 TB/s |",
          "|---|---|---|---|---|---|---|---|---|---|---|"]
for mode in KERNEL:
    d = durations(mode, "stats")
    if not d:
        continue
    t = d[-STEPS:]
    fe, wr = counters(mode, "fetch"), counters(mode, "write")
    alg = SAMPLES[mode] * BPS[mode]
    rd = 2.0 * fe["FETCH_SIZE"] * 1024 if "FETCH_SIZE" in fe else None
    ww = wr["WRITE_SIZE"] * 1024 if "WRITE_SIZE" in wr else None
    bj = os.path.join(src, f"bench_{mode}_unprofiled.json")
    bench = None
    if os.path.exists(bj):
        txt = [l for l in open(bj).read().splitlines() if l.startswith("{")]
        if txt:
            bench = json.loads(txt[-1])
            shutil.copy(bj, os.path.join(dst, f"{tag}_bench_{mode}.json"))
    if rd is not None and ww is not None:
        traffic[mode] = {"kernel": KERNEL[mode], "channels": C_, "frames": F_, "hbm_bytes_per_launch": int(rd + ww), "read_bytes": int(rd), "write_bytes": int(ww),
                         "source": f"profiles/{tag}_pmc_{mode}_fetch.csv + _write.csv (timed region, separate passes): FETCH_SIZE x2 x1024 + WRITE_SIZE x1024 "
                                   f"(MI355X_MICROARCH.md, HBM); recorded by tools/profile.sh, not measured in this run"}
    timed_rows.append({"mode": mode, "kernel": KERNEL[mode], "dispatches": len(t), "avg_ns": sum(t) / len(t), "min_ns": min(t), "max_ns": max(t),
                       "all_dispatches": len(d), "all_avg_ns": sum(d) / len(d)})
    lines.append(f"| {mode} | `{KERNEL[mode]}` | {sum(t) / len(t) / 1e3:.1f} | {min(t) / 1e3:.1f} | {max(t) / 1e3:.1f} | {alg / 1e6:.1f} | "
                 f"{'-' if rd is None else f'{rd / 1e6:.1f}'} | {'-' if ww is None else f'{ww / 1e6:.1f}'} | "
                 f"{'-' if rd is None or ww is None else f'{(rd + ww) / alg:.3f}'} | "
                 f"{'-' if not bench else bench['roofline']['kernel_avg_ms']} | {'-' if not bench else bench['roofline']['frac']} |")
lines.append("")
with open(os.path.join(dst, f"{tag}_kernel_timed.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(timed_rows[0].keys()) if timed_rows else ["mode"])
    w.writeheader()
    w.writerows(timed_rows)
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

for mode in ("meter", "roundtrip"):
    for sub in ("sq", "sq2"):
        c = counters(mode, sub)
        if c:
            d = durations(mode, sub)[-STEPS:]
            lines += [f"## SQ counters, `{mode}` pass `{sub}` (avg per timed dispatch of `{KERNEL[mode]}`; {sum(d) / max(len(d), 1) / 1e3:.1f} us under the profiler)", ""]
            lines += [f"* {k} = {v:.4g}" for k, v in sorted(c.items())]
            lines.append("")

bj = os.path.join(src, "bench_meter_unprofiled.json")
if os.path.exists(bj):
    txt = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if txt:
        lines += ["## un-profiled headline line of the same build", "", "```json", json.dumps(json.loads(txt[-1]), indent=1), "```", ""]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines[:60]))
