#!/bin/bash
# same-box A/B of other_configs legs: the tree's library vs ab_libs/<name>.so   bash tools/ab_legs.sh lib_r3 leg1 [leg2 ...]
set -u
OLD=$1; shift
ARGS=""; for l in "$@"; do ARGS="$ARGS --only-config $l"; done
cp titanet_amd/libtitanet_amd.so /tmp/lib_new.so
for which in new old new old; do
  if [ $which = new ]; then cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so; else cp ab_libs/${OLD}.so titanet_amd/libtitanet_amd.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --median-steps 0 --no-ceiling $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', d['ms_per_step'], {k: v.get('ms_per_step') for k, v in d['other_configs'].items()})"
done
cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so
