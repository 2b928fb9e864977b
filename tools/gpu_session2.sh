set -x
mkdir -p gpurun_out
(timeout 300 tools/pgemm_harness 76800 1024 1024 2>&1) > gpurun_out/r05_pgemm_harness_1024.txt
(timeout 300 tools/pgemm_harness 76800 512 512 2>&1) > gpurun_out/r05_pgemm_harness_512.txt
(timeout 300 tools/dgrad_dw_harness 2>&1) > gpurun_out/r05_dgrad_dw_wgx.txt
(timeout 600 python -m pytest tests/test_model_sizes_gpu.py tests/test_mask_gpu.py tests/test_config3_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r05_pytest2.txt
(timeout 900 bash tools/ab_legs.sh lib_v11 m10_b256 l5_bf16_b256 m10_ragged_mel_specaug_masked 2>&1) > gpurun_out/r05_ab_legs1.txt
(timeout 1200 python tools/train_compare.py --sig 0.012 --steps 1200 --train 128 --held 32 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_train_compare_sweep2.txt
cat gpurun_out/r05_pgemm_harness_1024.txt gpurun_out/r05_pgemm_harness_512.txt gpurun_out/r05_dgrad_dw_wgx.txt gpurun_out/r05_pytest2.txt gpurun_out/r05_ab_legs1.txt gpurun_out/r05_train_compare_sweep2.txt
