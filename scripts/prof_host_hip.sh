#!/bin/bash
# rocprofv3 kernel + HIP API stats of host_tester (where does the HOST time of a query go?):
#   bash scripts/prof_host_hip.sh <tag> <host_tester args...>  ->  gpurun_out/<tag>_{kernel,hip_api}_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- $R/plan_amd/host_tester "$@" > $R/gpurun_out/prof_$tag.out 2> $R/gpurun_out/prof_$tag.err < /dev/null
for kind in kernel hip_api; do
  f=$(find $R/gpurun_out/prof_$tag -name "*${kind}_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${tag}_${kind}_stats.csv; echo "== $kind"; head -14 "$f" | cut -c1-150; fi
done
grep "took" $R/gpurun_out/prof_$tag.err | tail -3
find $R/gpurun_out/prof_$tag -name '*trace.csv' -size +20M -delete
