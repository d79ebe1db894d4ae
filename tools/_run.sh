rm -f gpurun_out/steps.log
tools/gpu_step.sh gpu_tests 900 python -m pytest tests -m gpu -x -q || exit 99
for n in 164 240 80 24; do
tools/gpu_step.sh bench_rt_n$n 300 python bench.py --mode roundtrip --frame-bytes $n --no-cpu-baseline --no-stream-calib --placement abi || exit 99
done
tail -4 gpurun_out/gpu_tests.log
for n in 164 240 80 24; do tail -1 gpurun_out/bench_rt_n$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print($n, r['kernel'], r['kernel_avg_ms'], r['frac'], r['achieved'])"; done
