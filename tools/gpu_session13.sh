set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_trained_parity_gpu.py tests/test_model_sizes_gpu.py tests/test_mask_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/r05_pytest13.txt
(timeout 900 bash tools/ab_legs.sh lib_v11 l5_fp8_b256 2>&1) > gpurun_out/r05_ab_legs7.txt
bash tools/prof_leg.sh r05_l5_fp8_b256 l5_fp8_b256 > gpurun_out/r05_l5_fp8_b256_summary.txt 2>&1
cat gpurun_out/r05_pytest13.txt gpurun_out/r05_ab_legs7.txt; head -14 gpurun_out/r05_l5_fp8_b256_summary.txt | cut -c1-160
