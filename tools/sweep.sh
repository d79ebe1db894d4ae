#!/bin/bash
# frames sweep: separates fixed launch cost from per-chunk cost
for f in 8 32 128 256; do
  python3 /root/repo/bench.py --steps 10 --warmup 2 --frames $f --no-cpu-baseline --no-stream-calib | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames',d['config']['frames_per_launch'],'kernel_ms',d['roofline']['kernel_avg_ms'],'GB/s',d['roofline']['achieved'])"
done
