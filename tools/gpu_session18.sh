set -x
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -8) > gpurun_out/r05_pytest18.txt
cat gpurun_out/r05_pytest18.txt
(timeout 900 python tools/fuzz_paths.py 24 11 2>&1 | grep -E "^(ok|FAIL|worst)" | cut -c1-60,250-420) > gpurun_out/r05_fuzz18.txt
grep -c "^ok" gpurun_out/r05_fuzz18.txt; grep "^FAIL\|^worst" gpurun_out/r05_fuzz18.txt
bash tools/ab_env.sh TN_SE_FUSED 0 1 2 > gpurun_out/r05_ab_se_fused.txt 2>&1; cat gpurun_out/r05_ab_se_fused.txt
