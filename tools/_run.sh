rm -f gpurun_out/steps.log
tools/gpu_step.sh bench_wav84 300 python bench.py --mode wav --no-cpu-baseline --no-stream-calib --placement abi || exit 99
tools/gpu_step.sh bench_wav0 300 python bench.py --mode wav --no-cpu-baseline --no-stream-calib --placement abi --wav-offset 0 || exit 99
for x in 84 0; do tail -1 gpurun_out/bench_wav$x.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print($x, r['kernel'], r['kernel_avg_ms'], r['frac'], r['achieved'], d['config']['placement']['io_alloc_report']['classes_found'])"; done
