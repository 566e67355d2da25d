#!/bin/bash
# kernel-level timing of the aggregate sink at several cardinalities
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/aggprof_$c -o p -- python3 $R/scripts/agg_probe.py $c > $R/gpurun_out/aggprof_$c.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/aggprof_$c -name '*kernel_stats.csv' | head -1)
  echo "== card $c"; if [ -n "$f" ]; then cut -d, -f1-4,6 "$f" | head -5; else tail -3 $R/gpurun_out/aggprof_$c.log; fi
done
