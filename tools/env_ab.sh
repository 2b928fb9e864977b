#!/bin/bash
# same-box A/B of ONE other_configs leg under ENV=v for each listed value:  bash tools/env_ab.sh LEG ENV ROUNDS v1 v2 ...
LEG=$1; ENV=$2; R=$3; shift 3
for i in $(seq $R); do
for v in "$@"; do
env $ENV=$v python -c "
import sys, json, torch
sys.path.insert(0, '.')
import bench
torch.set_num_threads(8)
r = bench.other_configs(torch.device('cuda', 0), only=['$LEG'])
print('$LEG $ENV=$v', r['$LEG']['ms_per_step'])
" 2>&1 | grep "$ENV="
done; done
