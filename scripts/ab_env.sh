#!/bin/bash
# same-box A/B of one environment switch over host_tester queries: bash scripts/ab_env.sh VAR=VALUE q1 q2 ...   (best of 6 runs, with and without the switch)
R=${GRAFT_REPO_ROOT:-.}
sw=$1; shift
for q in "$@"; do
  a=$(timeout -k 10 120 $R/plan_amd/host_tester tpch $q 10 1 6 2>&1 | grep took | sed 's/.*took \([0-9.]*\)ms.*/\1/' | sort -n | head -1)
  b=$(env $sw timeout -k 10 120 $R/plan_amd/host_tester tpch $q 10 1 6 2>&1 | grep took | sed 's/.*took \([0-9.]*\)ms.*/\1/' | sort -n | head -1)
  echo "Q$q default $a ms, $sw $b ms"
done
