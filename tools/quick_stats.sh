#!/bin/bash
# kernel-time summary of one bench run (GPU box):  bash tools/quick_stats.sh TAG [bench args...]
set -u
TAG=${1:-q}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- \
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ceiling --median-steps 0 "$@" > gpurun_out/${TAG}_bench_profiled.json 2> gpurun_out/${TAG}_trace.log
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_trace
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
steps=29
for r in rows[:32]:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{r['Name'][:90]:90s} {n:5d} {t/n/1e3:9.1f}us {t/steps/1e3:9.1f}us/step")
print("total us/step", sum(float(r['TotalDurationNs']) for r in rows)/steps/1e3)
PY
