cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/quick_stats_ml.sh r04b_m10 m 10 bf16 > /dev/null 2>&1
bash tools/quick_stats_ml.sh r04b_l5 l 5 bf16 > /dev/null 2>&1
bash tools/quick_stats_ml.sh r04b_l5_fp8 l 5 fp8 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04b_ragged_trace -o rg -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-ceiling --median-steps 0 --only-config m10_ragged_mel_specaug_masked > gpurun_out/r04b_ragged.log 2>&1
find gpurun_out/r04b_ragged_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04b_ragged_kernel_stats.csv \;
rm -rf gpurun_out/r04b_ragged_trace
ls -la gpurun_out/r04b_*
