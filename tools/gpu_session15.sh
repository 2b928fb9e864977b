set -x
mkdir -p gpurun_out
(timeout 900 python tools/fuzz_paths.py 14 3 2>&1 | grep -v amdgpu | tail -18) > gpurun_out/r05_fuzz15.txt
cat gpurun_out/r05_fuzz15.txt
(timeout 1200 python -m pytest tests -m gpu -q -x -k "parity or golden or boundary or trained or shapes or sizes or masked" 2>&1 | grep -v amdgpu | tail -8) > gpurun_out/r05_pytest15.txt
cat gpurun_out/r05_pytest15.txt
bash tools/ab_env.sh TN_ASP_FUSED 0 1 2 > gpurun_out/r05_ab_asp1.txt 2>&1
cat gpurun_out/r05_ab_asp1.txt
