set -x
mkdir -p gpurun_out
for i in 1 2 3; do (timeout 300 python -m pytest "tests/test_trained_parity_gpu.py::test_gradients_at_trained_weights_vs_float64_oracle" -x -q -s -k "512-7-bf16" 2>&1 | grep -v amdgpu | grep "H=512\|assert\|Error\|passed\|failed" | head -8); done > gpurun_out/r05_dbg_e.txt
for i in 1 2; do (TN_KROT=0 timeout 300 python -m pytest "tests/test_trained_parity_gpu.py::test_gradients_at_trained_weights_vs_float64_oracle" -x -q -s -k "512-7-bf16" 2>&1 | grep -v amdgpu | grep "H=512\|assert\|Error\|passed\|failed" | head -8); done > gpurun_out/r05_dbg_f.txt
(timeout 1500 python -m pytest tests -m gpu -q --durations=15 2>&1 | grep -v amdgpu | tail -40) > gpurun_out/r05_pytest7.txt
cat gpurun_out/r05_dbg_e.txt gpurun_out/r05_dbg_f.txt gpurun_out/r05_pytest7.txt
