#!/bin/bash
# rocprofv3 --kernel-trace --stats of every secondary bench mode (no counters); one CSV per mode under gpurun_out/prof_modes/
out=/root/repo/gpurun_out/prof_modes
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in store roundtrip rtp packets depayload encode; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$m -- python3 /root/repo/bench.py --mode $m --steps 20 --warmup 3 --no-cpu-baseline > $out/$m.log 2>&1 || echo "$m failed"
done
find $out -name "*kernel_stats.csv" | head
