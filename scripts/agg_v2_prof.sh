#!/bin/bash
# kernel breakdown of the bulk build's second form at 65 k groups, per scatter shape: bash scripts/agg_v2_prof.sh [bu:tpb ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  bu=${cfg%%:*}; tpb=${cfg##*:}
  export PH_AGG_BULK_BU=$bu PH_AGG_BULK_TPB=$tpb
  rm -rf $R/gpurun_out/aggv2_${bu}_$tpb
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/aggv2_${bu}_$tpb -o p -- python3 $R/scripts/agg_probe.py 65536 > $R/gpurun_out/aggv2_${bu}_$tpb.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/aggv2_${bu}_$tpb -name '*kernel_stats.csv' | head -1)
  echo "== BU $bu TPB $tpb"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['AverageNs']) > 5000: print('  %-62s %3s x %8.1f us' % (r['Name'][:62], r['Calls'], float(r['AverageNs']) / 1000))
PY
done
