#!/bin/bash
# SF1 (204 MB, Infinity-Cache sized) A/B: cache policy and grid for the Q1 kernel
for nt in 1 0; do for grid in 256 512 1024; do
  r=$(PH_SCAN_NT=$nt PH_SCAN_GRID=$grid python bench.py --sf 1 --steps 200 --warmup 20 --no-cpu-baseline --no-q3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['achieved'],1), round(d['roofline']['min_launch_ms'],4), round(d['roofline']['avg_launch_ms'],4), round(d['ms_per_step'],4))")
  echo "q1 sf1 nt=$nt grid=$grid -> GB/s,min_ms,avg_ms,ms_per_step: $r"
done; done
