#!/bin/bash
# rocprofv3 kernel stats of host_tester (the C++ operator layer): bash scripts/prof_host.sh <tag> <repeat> <host_tester args...>
# e.g. bash scripts/prof_host.sh r3_q3_res 20 q3 10 1 resident 20   ->  gpurun_out/<tag>_kernel_stats.csv + per-query table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; steps=$2; shift; shift
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- $R/plan_amd/host_tester "$@" > $R/gpurun_out/prof_$tag.out 2> $R/gpurun_out/prof_$tag.err < /dev/null
f=$(find $R/gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv; python3 $R/scripts/kstats.py "$f" $steps; else tail -5 $R/gpurun_out/prof_$tag.err; fi
