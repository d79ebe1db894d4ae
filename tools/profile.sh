#!/bin/bash
# Round profile (run on the GPU box through gpurun): tools/profile.sh <tag> [modes...]
#   headline (bench.py default): rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | two SQ passes
#   every other mode           : --kernel-trace | --pmc FETCH_SIZE | --pmc WRITE_SIZE      (separate passes: TCC has 4 slots)
# Counters of one hardware block per pass only (tools/pmc_mode.sh explains why); every profiled run sits under a timeout.
# Outputs land under gpurun_out/prof_<tag>/<mode>/<pass>/ ; tools/summarize_profile.py <tag> turns them into profiles/.
tag=${1:-r03}; shift
modes=${@:-"meter store roundtrip depayload rtp packets window encode wav meter164 store164 roundtrip164 meter24"}
root=/root/repo/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
run() {  # run <mode> <pass> <rocprof args...>
    local mode=$1 pass=$2; shift 2
    local out=$root/$mode/$pass
    mkdir -p $out
    local margs="--mode $mode"
    case $mode in *164) margs="--mode ${mode%164} --frame-bytes 164";; *24) margs="--mode ${mode%24} --frame-bytes 24";; esac
    timeout -k 10 200 rocprofv3 "$@" --output-format csv -d $out -- python3 /root/repo/bench.py $margs --steps 20 --warmup 3 --no-cpu-baseline --no-stream-calib --placement abi > $out.log 2>&1
    local rc=$?
    echo "$mode/$pass rc=$rc"
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "profiled run timed out: stopping"; exit 99; }
    return 0
}
for m in $modes; do
    if [ $m = meter ]; then run $m stats --kernel-trace --stats; else run $m stats --kernel-trace; fi
    run $m fetch --kernel-trace --pmc FETCH_SIZE
    run $m write --kernel-trace --pmc WRITE_SIZE
    if [ $m = meter ] || [ $m = roundtrip ]; then
        run $m sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
        run $m sq2 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
    fi
done
# the un-profiled lines of the same build (only of the modes asked for)
for m in $modes; do
    case $m in
        meter) python3 /root/repo/bench.py > $root/bench_meter_unprofiled.json 2> $root/bench_meter_unprofiled.err;;
        *164)  python3 /root/repo/bench.py --mode ${m%164} --frame-bytes 164 --no-cpu-baseline > $root/bench_${m}_unprofiled.json 2> $root/bench_${m}_unprofiled.err;;
        *24)   python3 /root/repo/bench.py --mode ${m%24} --frame-bytes 24 --no-cpu-baseline > $root/bench_${m}_unprofiled.json 2> $root/bench_${m}_unprofiled.err;;
        *)     python3 /root/repo/bench.py --mode $m --no-cpu-baseline > $root/bench_${m}_unprofiled.json 2> $root/bench_${m}_unprofiled.err;;
    esac
done
ls $root
