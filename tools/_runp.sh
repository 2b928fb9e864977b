cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_y -o y -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_y.log 2>&1
python tools/prof_summary.py gpurun_out/prof_y/y_results.db | grep -E "wide|gemm_nt|TOTAL"
timeout 200 python bench.py --no-cpu-baseline --steps 30 2>&1 | tail -1 | cut -c1-330
