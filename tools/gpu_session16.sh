set -x
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_asp_fused_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu | tail -25)
