#!/bin/bash
# rocprofv3 kernel stats of one bench query (round 2): bash scripts/prof_r2.sh q3|q9|q1|q6 [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
q=$1; tag=${2:-r02}
rm -rf $R/gpurun_out/prof_$q
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$q -o p -- python3 $R/bench.py --query $q --steps 20 --warmup 3 --no-cpu-baseline --no-companions > $R/gpurun_out/prof_$q.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/prof_$q -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${tag}_${q}_sf10_kernel_stats.csv; cut -d, -f1-4 "$f" | head -40; else tail -5 $R/gpurun_out/prof_$q.log; fi
