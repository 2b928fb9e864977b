#!/bin/bash
# Round profile set for bench.py (run on the GPU box through gpurun; outputs under gpurun_out/, the summaries
# are then copied into profiles/).  Counter passes are separate runs, each with --pmc only.
#   usage: bash tools/collect_profiles.sh r01_v3
set -u
TAG=${1:-r01_v3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- \
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --median-steps 0 > gpurun_out/${TAG}_bench_profiled.json 2> gpurun_out/${TAG}_trace.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -o f -- \
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-ceiling --no-other-configs --median-steps 0 > /dev/null 2> gpurun_out/${TAG}_fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -o w -- \
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-ceiling --no-other-configs --median-steps 0 > /dev/null 2> gpurun_out/${TAG}_write.log
python tools/pmc_summary.py gpurun_out/${TAG}_pmc_traffic.json --steps 9 gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write | head -20
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -2
# un-profiled run last (this is the line the round reports); the fresh counter summary is put where bench.py looks for it
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2>/dev/null
tail -c 1500 gpurun_out/${TAG}_bench.json
# keep the merge-back small: only summaries travel
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
