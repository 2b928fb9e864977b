set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_mask_gpu.py tests/test_config3_gpu.py tests/test_model_sizes_gpu.py tests/test_fp8_gpu.py tests/test_v2_shapes_gpu.py tests/test_trained_parity_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r05_pytest8.txt
(timeout 600 python tools/fuzz_paths.py 60 2>&1 | tail -6) > gpurun_out/r05_fuzz8.txt
(timeout 900 bash tools/ab_legs.sh lib_v11 m10_ragged_mel_specaug_masked s17_padded_masked_b256 2>&1) > gpurun_out/r05_ab_legs3.txt
(timeout 600 python -m pytest "tests/test_train_compare_gpu.py" -x -q -s -k "m-10" 2>&1 | grep -v amdgpu | tail -14) > gpurun_out/r05_pytest8b.txt
cat gpurun_out/r05_pytest8.txt gpurun_out/r05_fuzz8.txt gpurun_out/r05_ab_legs3.txt gpurun_out/r05_pytest8b.txt
