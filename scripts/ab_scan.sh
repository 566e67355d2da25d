#!/bin/bash
# A/B of the fused scan kernels' tuning knobs (nt loads, grid size) on the SF10 bench.
for q in q1 q6; do
for nt in 1 0; do
for grid in 0 256 512 768 1024 2048 4096; do
  if [ "$q" = q1 ] && [ $grid -gt 1024 ]; then continue; fi
  if [ $grid = 0 ]; then unset PH_SCAN_GRID; else export PH_SCAN_GRID=$grid; fi
  r=$(PH_SCAN_NT=$nt python bench.py --query $q --steps 30 --warmup 5 --no-cpu-baseline --no-q3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['achieved'],1), round(d['roofline']['min_launch_ms'],4), round(d['roofline']['avg_launch_ms'],4))")
  echo "$q nt=$nt grid=$grid -> GB/s,min_ms,avg_ms: $r"
done; done; done
