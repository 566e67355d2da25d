#!/bin/bash
# HBM traffic (PMC) of the bench queries and of the aggregate sink: separate rocprofv3 passes for
# FETCH_SIZE and WRITE_SIZE (they do not fit one pass), kernel-trace only beside them.
#   bash scripts/pmc_r2.sh <tag>      -> gpurun_out/<tag>_pmc_summary.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r02}
rm -rf $R/gpurun_out/pmc_$tag; mkdir -p $R/gpurun_out/pmc_$tag
for q in q1 q6 q3 q9; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag/${q}_$c -o p -- python3 $R/bench.py --query $q --steps 5 --warmup 1 --no-cpu-baseline --no-companions > $R/gpurun_out/pmc_$tag/${q}_$c.log 2>&1 < /dev/null
    echo "$q $c done"
  done
done
for card in 4 175 65536; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag/agg${card}_$c -o p -- python3 $R/scripts/agg_probe.py $card > $R/gpurun_out/pmc_$tag/agg${card}_$c.log 2>&1 < /dev/null
    echo "agg $card $c done"
  done
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_$tag $R/gpurun_out/${tag}_pmc_summary.json
