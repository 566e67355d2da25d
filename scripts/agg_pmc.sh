#!/bin/bash
# PMC counters of the aggregate sink kernel: bash scripts/agg_pmc.sh <cardinality>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
c=$1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/aggpmc_$c/p$i -o p -- python3 $R/scripts/agg_probe.py $c > $R/gpurun_out/aggpmc_$c.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/aggpmc_$c/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if 'agg_sink' in row['Kernel_Name']:
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k, v in acc.items():
    print(f"{k}: last={v[-1]:.4g} n={len(v)}")
PY
  else tail -3 $R/gpurun_out/aggpmc_$c.log; fi
done
