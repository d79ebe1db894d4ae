rm -f gpurun_out/steps.log
tools/gpu_step.sh gpu_tests 900 python -m pytest tests -m gpu -x -q || exit 99
tail -4 gpurun_out/gpu_tests.log
