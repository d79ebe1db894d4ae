#!/bin/bash
# every bench mode once (pre-warm + output placement on), one JSON per mode under gpurun_out/modes/
mkdir -p gpurun_out/modes
for m in ${MODES:-meter store roundtrip rtp packets depayload encode}; do
  python3 bench.py --mode $m --no-cpu-baseline --steps 100 > gpurun_out/modes/$m.json 2> gpurun_out/modes/$m.err || { echo "$m failed"; tail -5 gpurun_out/modes/$m.err; continue; }
  python3 -c "import json; d=json.load(open('gpurun_out/modes/$m.json')); r=d['roofline']; p=d['config'].get('output_placement',{}); print('$m', r['kernel'], r['kernel_avg_ms'], r['achieved'], r['frac'], d['value'], p.get('kept'), p.get('straddle'), min(p.get('positions_ms',[0])))"
done
