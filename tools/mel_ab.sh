for i in 1 2; do
for g in 0 1; do
TN_MEL_GENERIC=$g python -c "
import sys, json, torch
sys.path.insert(0, '.')
import bench
torch.set_num_threads(8)
r = bench.other_configs(torch.device('cuda', 0), only=['m10_ragged_mel_specaug_masked'])
print('generic=$g', r['m10_ragged_mel_specaug_masked']['ms_per_step'])
" 2>&1 | grep generic
done; done
