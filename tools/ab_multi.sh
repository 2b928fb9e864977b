#!/bin/bash
# same-box comparison of SEVERAL libraries on the headline step, round-robin:  bash tools/ab_multi.sh ROUNDS lib_a lib_b ...   (names under ab_libs/, "tree" = the tree's)
set -u
ROUNDS=$1; shift
cp titanet_amd/libtitanet_amd.so /tmp/lib_tree.so
for r in $(seq 1 $ROUNDS); do
  for l in "$@"; do
    if [ $l = tree ]; then cp /tmp/lib_tree.so titanet_amd/libtitanet_amd.so; else cp ab_libs/${l}.so titanet_amd/libtitanet_amd.so; fi
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-ceiling --median-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$l', d['ms_per_step'], d['config'].get('params_finite'))"
  done
done
cp /tmp/lib_tree.so titanet_amd/libtitanet_amd.so
