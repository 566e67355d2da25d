#!/bin/bash
for u in 1 2 3; do
for grid in 128 192 256 320 384 512; do
  r=$(PH_SCAN_UNROLL=$u PH_SCAN_GRID=$grid python bench.py --query q1 --steps 30 --warmup 5 --no-cpu-baseline --no-q3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['achieved'],1), round(d['roofline']['min_launch_ms'],4), round(d['roofline']['avg_launch_ms'],4))")
  echo "q1 unroll=$u grid=$grid -> GB/s,min_ms,avg_ms: $r"
done; done
for grid in 128 192 256 320 384; do
  r=$(PH_SCAN_GRID=$grid python bench.py --query q6 --steps 30 --warmup 5 --no-cpu-baseline --no-q3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['achieved'],1), round(d['roofline']['min_launch_ms'],4), round(d['roofline']['avg_launch_ms'],4))")
  echo "q6 grid=$grid -> GB/s,min_ms,avg_ms: $r"
done
