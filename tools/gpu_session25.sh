set -x
mkdir -p gpurun_out
bash tools/collect_profiles.sh r05_v5 > gpurun_out/r05_v5_collect.log 2>&1
tail -3 gpurun_out/r05_v5_collect.log | cut -c1-300
for leg in m10_b256 m10_ragged_mel_specaug_masked; do bash tools/prof_leg.sh r05_$leg $leg > gpurun_out/r05_${leg}_summary.txt 2>&1; done
(timeout 900 python -m pytest tests -m gpu -q -x -k "eval or train_gpu or dp or asp or se_fused" 2>&1 | grep -v amdgpu | tail -3) > gpurun_out/r05_pytest25.txt
cat gpurun_out/r05_pytest25.txt
