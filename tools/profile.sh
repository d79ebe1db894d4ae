#!/bin/bash
# Round profile of the headline bench (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats                (per-kernel time; no counters)
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE       (own pass)
#   3. rocprofv3 --kernel-trace --pmc WRITE_SIZE       (own pass)
#   4. rocprofv3 --kernel-trace --pmc SQ_* issue/wait counters
# Outputs land under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/.
tag=${1:-r01}
out=/root/repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $out/sq -- $B > $out/sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq2 -- $B > $out/sq2.log 2>&1 || exit 1
python3 /root/repo/bench.py --steps 50 --warmup 5 > $out/bench_unprofiled.json 2> $out/bench_unprofiled.err
find $out -name "*.csv" | head -30
