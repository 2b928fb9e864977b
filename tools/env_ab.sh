#!/bin/bash
# same-box A/B of ONE other_configs leg under ENV=A / ENV=B:  bash tools/env_ab.sh LEG ENV A B [ROUNDS]
LEG=$1; ENV=$2; A=$3; B=$4; R=${5:-2}
for i in $(seq $R); do
for v in $A $B; do
env $ENV=$v python -c "
import sys, json, torch
sys.path.insert(0, '.')
import bench
torch.set_num_threads(8)
r = bench.other_configs(torch.device('cuda', 0), only=['$LEG'])
print('$ENV=$v', r['$LEG']['ms_per_step'])
" 2>&1 | grep "$ENV="
done; done
