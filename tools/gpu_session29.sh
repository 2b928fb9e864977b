set -x
mkdir -p gpurun_out
# same-box A/B: the library at the start of this session (commit 7bdcf39) vs the tree's, headline + wide legs, alternating
cp titanet_amd/libtitanet_amd.so /tmp/lib_new.so
for which in new old new old new old; do
  if [ $which = new ]; then cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so; else cp ab_libs/r05_start.so titanet_amd/libtitanet_amd.so; fi
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-ceiling --median-steps 100 --only-config m10_b256 --only-config m10_ragged_mel_specaug_masked --only-config l5_bf16_b256 --only-config l5_fp8_b256 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', d['ms_per_step'], 'median', d['roofline']['step_time_events']['median_ms'], {k: v.get('ms_per_step') for k, v in d['other_configs'].items()})"
done > gpurun_out/r05_ab_session.txt 2>&1
cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so
grep "^new\|^old" gpurun_out/r05_ab_session.txt
