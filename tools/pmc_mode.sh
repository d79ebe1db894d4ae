#!/bin/bash
# usage: tools/pmc_mode.sh <mode> <outdir-name> <counters...>
# One rocprofv3 --pmc pass (kernel-trace only) of `bench.py --mode <mode>`; prints the per-kernel averages of every counter.
#
# What can go wrong, and what this script does about it.  rocprofv3 builds its counter configuration inside the profiled
# process's FIRST HIP call; a set the hardware cannot collect in one pass is rejected there with "error code 38: Request exceeds
# the capabilities of the hardware to collect" and the process aborts (signal 6) before any kernel of ours has been dispatched
# (round 1: a TA_* + TCC_* set; round 2: a TA-only set, gpurun_out/pmc_packets_ta.log — so the limit is PER BLOCK, not about
# mixing blocks: every block has a fixed number of counter slots, derived counters such as X_sum expand to their base counters,
# and `rocprofv3 -L` lists blocks and dimensions but not the slot counts).  Therefore:
#   1. the requested set is written at the top of the log (round 2's abort could not be traced to its counter list);
#   2. counters of several blocks in one pass are still refused (one block per pass keeps the rule simple), except GRBM beside
#      SQ / TCC, which has always worked;
#   3. the set is PRE-FLIGHTED on a trivial HIP program (tools/pmc_preflight.sh: one torch fill kernel, 60 s limit): if the
#      profiler rejects it there, this script stops without having started bench.py;
#   4. the profiled bench run sits under its own timeout.
# tools/pmc_preflight.sh --limits prints how many counters of a block the profiler accepts in one pass (TA: see DESIGN.md 9).
mode=$1; name=$2; shift 2
blocks=$(for c in "$@"; do case $c in FETCH_SIZE|WRITE_SIZE) echo TCC;; *) echo "${c%%_*}";; esac; done | sort -u | tr '\n' ' ')
if [ "$(echo $blocks | wc -w)" -gt 1 ]; then
    case "$blocks" in "GRBM SQ "|"GRBM TCC ") ;; *) echo "pmc_mode.sh: counters of several blocks in one pass ($blocks): split them" >&2; exit 2;; esac
fi
# counter slots per block and pass, measured with tools/pmc_preflight.sh --limits on MI355X / ROCm 7.2 (profiles/r03_pmc_block_limits.txt):
# TA 2, TCP 4; TCC 4 and SQ 8 are the sets the round profiles have always used.  More than that is refused here already.
for blk in $blocks; do
    n=$(for c in "$@"; do case $c in FETCH_SIZE|WRITE_SIZE) echo TCC;; *) echo "${c%%_*}";; esac; done | grep -c "^$blk$")
    case $blk in TA) max=2;; TCP|TCC) max=4;; SQ) max=8;; *) max=8;; esac
    if [ "$n" -gt "$max" ]; then echo "pmc_mode.sh: $n $blk counters in one pass, the block takes $max: split them" >&2; exit 2; fi
done
out=/root/repo/gpurun_out/pmc_$name
mkdir -p $out
{ echo "# pmc_mode.sh $(date -u +%FT%TZ) mode=$mode blocks=[$blocks] counters: $*"; } > "$out.log"
bash "$(dirname "$0")/pmc_preflight.sh" "$@" >> "$out.log" 2>&1
pf=$?
if [ $pf -ne 0 ]; then echo "pmc_mode.sh: the profiler rejects this counter set (pre-flight rc=$pf, see $out.log): not starting bench.py" >&2; exit 3; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -- python3 /root/repo/bench.py --mode $mode --steps 5 --warmup 2 --no-cpu-baseline --prewarm-ms 0 --placement abi --no-stream-calib >> "$out.log" 2>&1
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
