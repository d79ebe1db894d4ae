rm -f gpurun_out/steps.log
tools/gpu_step.sh gpu_tests 900 python -m pytest tests -m gpu -x -q || exit 99
for i in 1 2 3; do tools/gpu_step.sh bench_meter$i 300 python bench.py --no-cpu-baseline --no-stream-calib || exit 99; done
tail -3 gpurun_out/gpu_tests.log
for i in 1 2 3; do tail -1 gpurun_out/bench_meter$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_avg_ms'], r['frac'], r['frac_unassisted'], d['config']['placement']['io_alloc_report'])"; done
