#!/bin/bash
# same-box A/B of the fused scan tail (last-workgroup merge + publish) against separate merge / publish launches:
#   bash scripts/ab_scan_tail.sh      -> "Query N took" minima per variant, two rounds each
R=${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
  for q in 1 6; do
    for tail in 1 0; do
      m=$(PH_SCAN_TAIL=$tail timeout -k 10 120 $R/plan_amd/host_tester tpch $q 10 1 12 2>&1 | grep took | sed 's/.*took \([0-9.]*\)ms.*/\1/' | sort -n | head -3 | tr '\n' ' ')
      echo "round $round Q$q PH_SCAN_TAIL=$tail: best three $m"
    done
  done
done
