#!/bin/bash
# usage: tools/pmc_mode.sh <mode> <outdir-name> <counters...>
# One rocprofv3 --pmc pass (kernel-trace only) of `bench.py --mode <mode>`; prints the per-kernel averages of every counter.
#
# Counters of DIFFERENT hardware blocks must not share a pass: in round 1 a pass that mixed TA_* with TCC_* counters made
# rocprofv3 abort inside the first HIP call of the profiled process ("error code 38 ... exceeds the capabilities of the
# hardware", signal 6 in igdsp_create, before any igdsp kernel had been dispatched — gpurun_out/pmc_ta_*.log) and the
# half-dead process then sat until SIGTERM.  The profiler rejected the counter SET; no kernel of ours was involved.  So this
# script refuses mixed sets up front and runs the profiled program under its own timeout.
mode=$1; name=$2; shift 2
blocks=$(for c in "$@"; do case $c in FETCH_SIZE|WRITE_SIZE) echo TCC;; *) echo "${c%%_*}";; esac; done | sort -u | tr '\n' ' ')
if [ "$(echo $blocks | wc -w)" -gt 1 ]; then
    case "$blocks" in "GRBM SQ "|"GRBM TCC ") ;; *) echo "pmc_mode.sh: counters of several blocks in one pass ($blocks): split them" >&2; exit 2;; esac
fi
out=/root/repo/gpurun_out/pmc_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -- python3 /root/repo/bench.py --mode $mode --steps 5 --warmup 2 --no-cpu-baseline --prewarm-ms 0 --placement abi --no-stream-calib > "$out.log" 2>&1
rc=$?
if [ $rc -ne 0 ]; then echo "pmc_mode.sh: profiled run ended with rc=$rc (see $out.log)" >&2; tail -5 "$out.log" >&2; exit $rc; fi
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "igdsp" in k and "gen_uniform" not in k and "stream" not in k and "hold_reset" not in k:
        print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "dispatches", len(next(iter(c.values()))))
PY
