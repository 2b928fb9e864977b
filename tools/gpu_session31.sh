set -x
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x -k "backward or mask or sizes or config3 or trained or se_fused" 2>&1 | grep -v amdgpu | grep "passed\|failed\|FAILED" | tail -4) > gpurun_out/r05_pytest31.txt
cat gpurun_out/r05_pytest31.txt
(timeout 600 python tools/fuzz_paths.py 12 51 2>&1 | grep -E "^(ok|FAIL|worst)" | cut -c1-60,250-420) > gpurun_out/r05_fuzz31.txt
grep -c "^ok" gpurun_out/r05_fuzz31.txt; grep "^FAIL\|^worst" gpurun_out/r05_fuzz31.txt
cp titanet_amd/libtitanet_amd.so /tmp/lib_new.so
for which in new old new old new old; do
  if [ $which = new ]; then cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so; else cp ab_libs/head5.so titanet_amd/libtitanet_amd.so; fi
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-ceiling --median-steps 100 --only-config m10_b256 --only-config l5_bf16_b256 --only-config m10_ragged_mel_specaug_masked 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', d['ms_per_step'], 'median', d['roofline']['step_time_events']['median_ms'], {k: v.get('ms_per_step') for k, v in d['other_configs'].items()})"
done > gpurun_out/r05_ab_cb1.txt 2>&1
cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so
grep "^new\|^old" gpurun_out/r05_ab_cb1.txt
