set -x
for leg in m10_ragged_mel_specaug_masked l5_fp8_b256 m10_b256 l5_bf16_b256; do
  bash tools/prof_leg.sh r05_${leg} $leg > gpurun_out/r05_${leg}_summary.txt 2>&1
done
head -45 gpurun_out/r05_m10_ragged_mel_specaug_masked_summary.txt; head -40 gpurun_out/r05_l5_fp8_b256_summary.txt
