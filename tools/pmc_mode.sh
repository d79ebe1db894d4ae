#!/bin/bash
# usage: tools/pmc_mode.sh <mode> <outdir-name> <counters...> : one rocprofv3 --pmc pass (kernel-trace only) of bench.py --mode <mode>
mode=$1; name=$2; shift 2
out=/root/repo/gpurun_out/pmc_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -- python3 /root/repo/bench.py --mode $mode --steps 5 --warmup 2 --no-cpu-baseline --prewarm-ms 0 --placement-positions 1 > "$out.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "igdsp" in k and "gen_uniform" not in k and "stream" not in k:
        print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "dispatches", len(next(iter(c.values()))))
PY
# note: rocprofv3 aborted (signal 6) on a pass that mixed TA_* and TCC_* counters; keep a pass to one block's counters (SQ_*, or TCC_*, ...)
