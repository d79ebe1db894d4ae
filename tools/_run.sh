rm -f gpurun_out/steps.log
for m in meter store roundtrip encode; do
IGDSP_IO_DEBUG=1 tools/gpu_step.sh io_place_$m 300 python tools/io_place.py $m || exit 99
done
for m in meter store roundtrip encode; do grep -v "igdsp_io\] chunk" gpurun_out/io_place_$m.log | grep -v "^{" | cut -c1-700; done
