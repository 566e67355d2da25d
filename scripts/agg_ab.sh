#!/bin/bash
# A/B of aggregate-sink variants on one box: bash scripts/agg_ab.sh "<env assignments>" card...
envs=$1; shift
for c in "$@"; do
  out=$(env $envs bash $GRAFT_REPO_ROOT/scripts/agg_prof.sh $c 2>&1 | grep "agg_sink\|bulk_" | cut -d, -f1,4 | tr '\n' ' ')
  echo "[$envs] card $c: $out"
done
