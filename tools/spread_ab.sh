for fl in "-DIGDSP_SPREAD_METER=0" "-DIGDSP_SPREAD_METER=1"; do
  IGDSP_CXXFLAGS="$fl" python3 -m igate4xsoftphonedsp_amd.build --force > /dev/null 2>&1 || { echo "build failed $fl"; continue; }
  for m in meter rtp packets; do
    for i in 1 2; do
      python3 bench.py --mode $m --no-cpu-baseline --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$fl $m', d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
    done
  done
done
python3 -m igate4xsoftphonedsp_amd.build --force > /dev/null 2>&1
