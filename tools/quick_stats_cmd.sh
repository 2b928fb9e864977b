#!/bin/bash
# kernel-time summary of an arbitrary command (GPU box):  bash tools/quick_stats_cmd.sh TAG STEPS cmd...
set -u
TAG=${1:-q}; STEPS=${2:-1}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- "$@" > gpurun_out/${TAG}.log 2> gpurun_out/${TAG}_trace.log
tail -2 gpurun_out/${TAG}.log | cut -c1-600
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_trace
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
steps=${STEPS}
for r in rows[:36]:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{r['Name'][:100]:100s} {n:5d} {t/n/1e3:9.1f}us {t/steps/1e3:9.1f}us/step")
print("total us/step", sum(float(r['TotalDurationNs']) for r in rows)/steps/1e3)
PY
