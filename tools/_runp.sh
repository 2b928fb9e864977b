cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 127; do
TN_V2=$v timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_q$v -o y -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_y.log 2>&1
echo "TN_V2=$v"; python tools/prof_summary.py gpurun_out/prof_q$v/y_results.db | grep -E "dgrad_v2|dw_bwd_v4|sub_bwd_v6|TOTAL"
done
