set -x
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v amdgpu | tail -25) > gpurun_out/r05_pytest11.txt
(timeout 900 bash tools/ab_legs.sh lib_v11 l5_bf16_b256 m10_b256 s17_arcface_b256 2>&1) > gpurun_out/r05_ab_legs6.txt
cat gpurun_out/r05_pytest11.txt gpurun_out/r05_ab_legs6.txt
