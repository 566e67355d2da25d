#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/aggapi
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/aggapi -o p -- python3 $R/scripts/agg_probe.py 175 > $R/gpurun_out/aggapi.log 2>&1 < /dev/null
for f in $(find $R/gpurun_out/aggapi -name '*hip_api_stats.csv' -o -name '*kernel_stats.csv'); do echo "== $f"; cut -d, -f1-4 $f | head -14; done
