set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_trained_parity_gpu.py -x -q -s -k "fp8" 2>&1 | grep -v amdgpu | grep "H=\|M/10\|passed\|failed\|Error\|assert\|cosine" | cut -c1-400 | tail -24) > gpurun_out/r05_pytest10.txt
(timeout 900 bash tools/ab_legs.sh lib_v11 l5_fp8_b256 m10_b256 2>&1) > gpurun_out/r05_ab_legs5.txt
(timeout 600 python tools/fuzz_paths.py 40 2>&1 | tail -3) > gpurun_out/r05_fuzz10.txt
cat gpurun_out/r05_pytest10.txt gpurun_out/r05_ab_legs5.txt gpurun_out/r05_fuzz10.txt
