set -x
mkdir -p gpurun_out
(timeout 300 tools/pgemm_harness 76800 512 512 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl" | head -16) > gpurun_out/r05_pgemm_harness_512_v4.txt
(timeout 600 python -m pytest tests/test_model_sizes_gpu.py tests/test_mask_gpu.py tests/test_config3_gpu.py tests/test_fp8_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r05_pytest4b.txt
(timeout 900 bash tools/ab_legs.sh lib_v11 m10_b256 l5_bf16_b256 l5_fp8_b256 m10_ragged_mel_specaug_masked 2>&1) > gpurun_out/r05_ab_legs2.txt
(timeout 1200 python -m pytest tests/test_train_compare_gpu.py tests/test_trained_parity_gpu.py -x -q --durations=12 2>&1 | grep -v amdgpu.ids | tail -30) > gpurun_out/r05_pytest4.txt
cat gpurun_out/r05_pgemm_harness_512_v4.txt gpurun_out/r05_pytest4b.txt gpurun_out/r05_ab_legs2.txt gpurun_out/r05_pytest4.txt
