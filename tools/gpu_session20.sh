set -x
mkdir -p gpurun_out
bash tools/collect_profiles.sh r05_v3 > gpurun_out/r05_v3_collect.log 2>&1
tail -12 gpurun_out/r05_v3_collect.log | cut -c1-900
