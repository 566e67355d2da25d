#!/bin/bash
# HBM traffic (PMC, one rocprofv3 pass per counter) of named kernels of a host_tester run:
#   bash scripts/pmc_host.sh <tag> "<name1|name2|...>" <host_tester args...>     -> gpurun_out/<tag>_pmc.txt  (KiB per dispatch: FETCH_SIZE x 2 = bytes read / 1024)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; names=$2; shift; shift
out=$R/gpurun_out/${tag}_pmc.txt
: > $out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$tag/$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag/$c -o p -- $R/plan_amd/host_tester "$@" > $R/gpurun_out/pmc_$tag.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/pmc_$tag/$c -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$names" >> $out <<'PY'
import csv, sys, collections
names = sys.argv[2].split('|')
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    for nm in names:
        if nm in row['Kernel_Name']:
            acc[(nm, row['Counter_Name'])].append(float(row['Counter_Value']))
for (nm, k), v in sorted(acc.items()):
    v = v[len(v) // 2:]   # the later dispatches (warm)
    print(f"{nm:36s} {k:12s} avg={sum(v) / len(v):.1f} KiB n={len(v)}")
PY
  else echo "no counters ($c)" >> $out; tail -3 $R/gpurun_out/pmc_$tag.log >> $out; fi
done
cat $out
