#!/bin/bash
# the default bench in N separate processes (process-to-process variation; placement probe on / off)
for i in 1 2 3; do
for pc in 7 1; do
python3 bench.py --no-cpu-baseline --placement-positions $pc $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cands=$pc', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['same_traffic_stream_ms'], d['config'].get('output_placement'))"
done
done
