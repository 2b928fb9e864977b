set -x
mkdir -p gpurun_out
bash tools/collect_profiles.sh r05_v8 > gpurun_out/r05_v8_collect.log 2>&1
tail -2 gpurun_out/r05_v8_collect.log | cut -c1-200
grep "cast_params\|swizzle" gpurun_out/r05_v8_kernel_stats.csv | cut -c1-200
