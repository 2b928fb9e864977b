set -x
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x -k "sizes or mask or config3 or fp8" 2>&1 | grep -v amdgpu | tail -4) > gpurun_out/r05_pytest22.txt
cat gpurun_out/r05_pytest22.txt
for r in 1 2; do for w in 256 512; do TN_DWF_WGS=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline --median-steps 0 --no-ceiling --only-config m10_b256 --only-config m10_ragged_mel_specaug_masked --only-config l5_fp8_b256 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('TN_DWF_WGS=$w', {k: v.get('ms_per_step') for k, v in d['other_configs'].items()})"; done; done > gpurun_out/r05_ab_dwf_wgs.txt 2>&1
cat gpurun_out/r05_ab_dwf_wgs.txt
