cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_v2_shapes_gpu.py tests/test_full_size_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head
for v in 31 63; do
TN_V2=$v timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_q$v -o y -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_y.log 2>&1
echo "TN_V2=$v"; python tools/prof_summary.py gpurun_out/prof_q$v/y_results.db | grep -E "wgrad_batched|sub_fwd_v5|TOTAL"
done
