#!/bin/bash
# A/B: rebuild with -D flags on the box and bench each (the diagnostic IGDSP_AB_* knobs — builds whose RESULTS ARE WRONG —
# live in tools/ab_knobs.patch: `git apply tools/ab_knobs.patch` first, `git apply -R` after); usage: tools/ab.sh "<flags A>" "<flags B>" ...  (bench args via BENCH_ARGS)
for fl in "$@"; do
  IGDSP_CXXFLAGS="$fl" python3 -m igate4xsoftphonedsp_amd.build --force > /dev/null 2>&1 || { echo "build failed: $fl"; continue; }
  for i in 1 2; do
    python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-stream-calib $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$fl] [$BENCH_ARGS]', d['roofline']['kernel_avg_ms'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
python3 -m igate4xsoftphonedsp_amd.build --force > /dev/null 2>&1
