rm -f gpurun_out/steps.log
for m in meter store roundtrip encode; do
IGDSP_IO_DEBUG=1 tools/gpu_step.sh io_place_$m 300 python tools/io_place.py $m || exit 99
done
for m in meter store roundtrip encode; do grep -v "igdsp_io\] chunk" gpurun_out/io_place_$m.log | grep -v "^{" | cut -c1-500; done
tools/gpu_step.sh bench_meter 300 python bench.py || exit 99
tools/gpu_step.sh bench_coll 300 python bench.py --force-collective --no-cpu-baseline || exit 99
tools/gpu_step.sh gpu_tests 900 python -m pytest tests -m gpu -x -q || exit 99
tail -1 gpurun_out/bench_meter.log | cut -c1-1500; tail -1 gpurun_out/bench_coll.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('collective'), d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_unassisted'])"; tail -3 gpurun_out/gpu_tests.log
