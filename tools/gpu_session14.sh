set -x
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -6) > gpurun_out/r05_pytest14.txt
bash tools/collect_profiles.sh r05_v1 > gpurun_out/r05_v1_collect.log 2>&1
cat gpurun_out/r05_pytest14.txt; tail -30 gpurun_out/r05_v1_collect.log | cut -c1-600
