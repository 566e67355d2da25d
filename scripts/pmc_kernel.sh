#!/bin/bash
# PMC counters of one kernel of a bench query: bash scripts/pmc_kernel.sh <query> <kernel substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
q=$1; kn=$2
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$q/p$i -o p -- python3 $R/bench.py --query $q --steps 3 --warmup 1 --no-cpu-baseline --no-companions > $R/gpurun_out/pmc_$q.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/pmc_$q/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$kn" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row['Kernel_Name']:
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k, v in acc.items():
    print(f"{k}: " + " ".join(f"{x:.4g}" for x in v[-2:]))
PY
  else tail -3 $R/gpurun_out/pmc_$q.log; fi
done
