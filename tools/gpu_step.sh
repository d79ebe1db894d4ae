#!/bin/bash
# usage: tools/gpu_step.sh <name> <timeout-seconds> <command...>   (on the GPU box)
# Runs one GPU step under its own timeout, output to gpurun_out/<name>.log; a step that timed out or was killed ends the
# whole call (exit 99), an ordinary failure only reports (so later steps still run).
name=$1; lim=$2; shift 2
mkdir -p gpurun_out
echo "== $name: $*" | tee -a gpurun_out/steps.log
timeout -k 10 "$lim" "$@" > "gpurun_out/$name.log" 2>&1
rc=$?
echo "== $name rc=$rc" | tee -a gpurun_out/steps.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping" | tee -a gpurun_out/steps.log; exit 99; fi
exit 0
