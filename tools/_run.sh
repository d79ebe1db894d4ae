rm -f gpurun_out/steps.log
bash tools/profile.sh r02 > gpurun_out/profile_r02.log 2>&1
tail -40 gpurun_out/profile_r02.log
