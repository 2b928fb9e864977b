#!/bin/bash
# same-box A/B of the headline step under an environment switch read at plan creation (GPU box)
#   bash tools/ab_env.sh TN_OVERLAP 0 1 [rounds] [bench args...]
set -u
VAR=$1; A=$2; B=$3; ROUNDS=${4:-2}; shift; shift; shift; shift
for r in $(seq 1 $ROUNDS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-ceiling --median-steps 100 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v', d['ms_per_step'], 'median', d['roofline']['step_time_events']['median_ms'], d['roofline'].get('class_ms_per_step'), d['config'].get('params_finite'))"
  done
done
