#!/bin/bash
# A/B: rebuild with -D flags on the box and bench each; usage: tools/ab.sh "<flags A>" "<flags B>" ...  (bench args via BENCH_ARGS).
# The knobs are the #ifdef / #ifndef switches the kernel sources carry themselves: -DIGDSP_NT_STORE=1, -DIGDSP_NO_SPREAD,
# -DIGDSP_SPREAD_METER=1, -DIGDSP_RT_WAVES=N, -DIGDSP_RTL_WAVES=N, the waves per CU of the write-heavy kernels (-DIGDSP_STORE_WAVES=N,
# -DIGDSP_SSTORE_WAVES=N, -DIGDSP_ENC_WAVES=N), -DIGDSP_TINY_LATE, -DIGDSP_BLK_NORUN / -DIGDSP_BLK_STAMP (tools/blk_stamp.py) and, for the fused window kernel, -DIGDSP_WIN_NOBOOK /
# -DIGDSP_WIN_ASC (those two produce WRONG windows on purpose: what the kernel's time does not depend on).  (Round 1's
# wrong-result IGDSP_AB_* knobs lived in a patch against a file that no longer exists; they are in git history.)
for fl in "$@"; do
  IGDSP_CXXFLAGS="$fl" python3 -m igate4xsoftphonedsp_amd.build --force > /dev/null 2>&1 || { echo "build failed: $fl"; continue; }
  for i in 1 2; do
    python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-stream-calib $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$fl] [$BENCH_ARGS]', d['roofline']['kernel_avg_ms'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
python3 -m igate4xsoftphonedsp_amd.build --force > /dev/null 2>&1
