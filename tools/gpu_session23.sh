set -x
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -6) > gpurun_out/r05_pytest23.txt
cat gpurun_out/r05_pytest23.txt
(timeout 900 python tools/fuzz_paths.py 24 31 2>&1 | grep -E "^(ok|FAIL|worst)" | cut -c1-60,250-420) > gpurun_out/r05_fuzz23.txt
grep -c "^ok" gpurun_out/r05_fuzz23.txt; grep "^FAIL\|^worst" gpurun_out/r05_fuzz23.txt
bash tools/collect_profiles.sh r05_v4 > gpurun_out/r05_v4_collect.log 2>&1
tail -3 gpurun_out/r05_v4_collect.log | cut -c1-600
for leg in m10_b256 l5_bf16_b256 l5_fp8_b256 m10_ragged_mel_specaug_masked; do bash tools/prof_leg.sh r05_$leg $leg > gpurun_out/r05_${leg}_summary.txt 2>&1; done
ls gpurun_out | grep summary
