rm -f gpurun_out/steps.log
tools/gpu_step.sh flush_latency 300 python tools/flush_latency.py || exit 99
tools/gpu_step.sh gpu_tests 900 python -m pytest tests -m gpu -x -q -k "staging or host" || exit 99
cat gpurun_out/flush_latency.log; tail -3 gpurun_out/gpu_tests.log
