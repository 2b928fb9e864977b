set -x
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x -k "sizes or mask or config3 or forward or fp8 or backward" 2>&1 | grep -v amdgpu | tail -4) > gpurun_out/r05_pytest21.txt
cat gpurun_out/r05_pytest21.txt
(timeout 900 python tools/fuzz_paths.py 16 23 2>&1 | grep -E "^(ok|FAIL|worst)" | cut -c1-60,250-420) > gpurun_out/r05_fuzz21.txt
grep -c "^ok" gpurun_out/r05_fuzz21.txt; grep "^FAIL\|^worst" gpurun_out/r05_fuzz21.txt
bash tools/ab_legs.sh head_sef m10_b256 l5_bf16_b256 m10_ragged_mel_specaug_masked > gpurun_out/r05_ab_legs9.txt 2>&1; cat gpurun_out/r05_ab_legs9.txt
