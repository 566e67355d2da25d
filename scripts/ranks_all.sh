#!/bin/bash
# every TPC-H query over N ranks (threads, in-process transport) against the SF1 goldens: bash scripts/ranks_all.sh <nranks>
R=${GRAFT_REPO_ROOT:-/root/repo}
n=${1:-2}
for q in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22; do
  timeout -k 5 120 $R/plan_amd/host_tester ranks $n $q 1 1 > /tmp/rk_$q.out 2> /tmp/rk_$q.err
  rc=$?
  if [ $rc -eq 0 ] && cmp -s /tmp/rk_$q.out $R/tests/golden/plan_q$q.txt; then echo "q$q ok"; else echo "q$q FAIL rc=$rc: $(grep -v '^Query\|^plan run\|^scan#\|^join#\|^agg#\|^  ' /tmp/rk_$q.err | tail -2 | tr '\n' ' ')"; fi
done
