set -x
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -6) > gpurun_out/r05_pytest17.txt
cat gpurun_out/r05_pytest17.txt
(timeout 900 python tools/fuzz_paths.py 24 11 2>&1 | grep -E "^(ok|FAIL|worst)" | cut -c1-60,250-420) > gpurun_out/r05_fuzz17.txt
grep -c "^ok" gpurun_out/r05_fuzz17.txt; grep "^FAIL\|^worst" gpurun_out/r05_fuzz17.txt
bash tools/collect_profiles.sh r05_v2 > gpurun_out/r05_v2_collect.log 2>&1
tail -12 gpurun_out/r05_v2_collect.log | cut -c1-700
