set -x
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x -k "sizes or mask or config3 or fp8 or forward" 2>&1 | grep -v amdgpu | tail -4) > gpurun_out/r05_pytest24.txt
cat gpurun_out/r05_pytest24.txt
bash tools/ab_legs.sh head2 m10_b256 l5_bf16_b256 m10_ragged_mel_specaug_masked > gpurun_out/r05_ab_legs10.txt 2>&1; cat gpurun_out/r05_ab_legs10.txt
