set -x
mkdir -p gpurun_out
(timeout 300 tools/pgemm_harness 76800 512 512 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl") > gpurun_out/r05_pgemm_harness_512_v3.txt
(timeout 300 tools/pgemm_harness_stamps 76800 512 512 2>&1 | grep "rwgemm\|stamps") > gpurun_out/r05_rwgemm_stamps.txt
(timeout 300 tools/pgemm_harness 76800 1024 1024 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl") > gpurun_out/r05_pgemm_harness_1024_b.txt
(timeout 900 python -m pytest tests/test_train_compare_gpu.py tests/test_trained_parity_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -60) > gpurun_out/r05_pytest3.txt
(timeout 600 python -m pytest tests/test_model_sizes_gpu.py tests/test_mask_gpu.py tests/test_config3_gpu.py tests/test_fp8_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r05_pytest3b.txt
cat gpurun_out/r05_pgemm_harness_512_v3.txt gpurun_out/r05_rwgemm_stamps.txt gpurun_out/r05_pgemm_harness_1024_b.txt gpurun_out/r05_pytest3.txt gpurun_out/r05_pytest3b.txt
