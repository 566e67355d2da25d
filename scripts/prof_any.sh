#!/bin/bash
# rocprofv3 kernel stats of any script: bash scripts/prof_any.sh <tag> <script.py> [args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/"$@" > $R/gpurun_out/prof_$tag.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:86]:86s} calls {r['Calls']:>5s}  avg us {float(r['AverageNs'])/1e3:9.1f}")
PY
else tail -5 $R/gpurun_out/prof_$tag.log; fi
