#!/bin/bash
# same-box A/B of the headline step: the tree's library vs ab_libs/<name>.so, alternating (GPU box)
#   bash tools/ab_bench.sh lib_r3 [rounds] [bench args...]
set -u
OLD=${1:-lib_r3}; ROUNDS=${2:-2}; shift; shift
cp titanet_amd/libtitanet_amd.so /tmp/lib_new.so
for r in $(seq 1 $ROUNDS); do
  for which in new old; do
    if [ $which = new ]; then cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so; else cp ab_libs/${OLD}.so titanet_amd/libtitanet_amd.so; fi
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-ceiling --median-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', d['ms_per_step'], d['roofline'].get('class_ms_per_step'), d.get('params_finite'))"
  done
done
cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so
