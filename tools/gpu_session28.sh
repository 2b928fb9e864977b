set -x
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x -k "forward or backward or parity or sizes or train_gpu or boundary or mask" 2>&1 | grep -v amdgpu | grep "passed\|failed\|FAILED\|Error" | tail -5) > gpurun_out/r05_pytest28.txt
cat gpurun_out/r05_pytest28.txt
bash tools/prof_headline.sh r05_tail "tail_|cast_params|adam|se_wgrad|head_" 2>&1 | tail -14
cp titanet_amd/libtitanet_amd.so /tmp/lib_new.so
for which in new old new old; do
  if [ $which = new ]; then cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so; else cp ab_libs/head4.so titanet_amd/libtitanet_amd.so; fi
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-ceiling --median-steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', d['ms_per_step'], 'median', d['roofline']['step_time_events']['median_ms'])"
done > gpurun_out/r05_ab_tail.txt 2>&1
cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so
grep "^new\|^old" gpurun_out/r05_ab_tail.txt
