#!/bin/bash
# PMC counters of the kernels whose name contains one of the given substrings, one rocprofv3 pass per counter set:
#   bash scripts/pmc_any.sh <tag> "<name1|name2|...>" <python script> [args...]     -> gpurun_out/<tag>_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; names=$2; shift; shift
out=$R/gpurun_out/${tag}_pmc.txt
: > $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_$tag/p$i
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -o p -- python3 $R/"$@" > $R/gpurun_out/pmc_$tag.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/pmc_$tag/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$names" >> $out <<'PY'
import csv, sys, collections
names = sys.argv[2].split('|')
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    for nm in names:
        if nm in row['Kernel_Name']:
            acc[(nm, row['Counter_Name'])].append(float(row['Counter_Value']))
for (nm, k), v in sorted(acc.items()):
    print(f"{nm:28s} {k:24s} last={v[-1]:.5g} n={len(v)}")
PY
  else echo "set $i: no counters ($set)" >> $out; tail -3 $R/gpurun_out/pmc_$tag.log >> $out; fi
done
cat $out
