rm -f gpurun_out/steps.log
tools/gpu_step.sh gpu_tests 900 python -m pytest tests -m gpu -x -q -s -k "io_alloc" || exit 99
tail -5 gpurun_out/gpu_tests.log | cut -c1-300
