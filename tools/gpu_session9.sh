set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -s 2>&1 | grep -v amdgpu | grep "H=\|passed\|failed\|Error\|assert" | tail -20) > gpurun_out/r05_pytest9.txt
(timeout 900 python -m pytest tests/test_trained_parity_gpu.py tests/test_model_sizes_gpu.py -x -q -s -k "fp8 or finite" 2>&1 | grep -v amdgpu | grep "H=\|M/10\|passed\|failed\|Error\|assert\|cosine" | tail -20) >> gpurun_out/r05_pytest9.txt
(timeout 900 bash tools/ab_legs.sh lib_v11 l5_fp8_b256 2>&1) > gpurun_out/r05_ab_legs4.txt
(timeout 900 python -m pytest tests/test_train_compare_gpu.py -x -q -s -k "fp8" 2>&1 | grep -v amdgpu | grep "loss_last\|acc_last\|eer:\|passed\|failed\|TitaNet" | tail -20) > gpurun_out/r05_pytest9b.txt
cat gpurun_out/r05_pytest9.txt gpurun_out/r05_ab_legs4.txt gpurun_out/r05_pytest9b.txt
