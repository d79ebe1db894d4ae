#!/bin/bash
# usage: tools_pmc.sh <outdir> <counters...>   (one rocprofv3 --pmc pass, kernel-trace only)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -- python3 /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out.log" 2>&1
