#!/bin/bash
# kernel-time summary of a few TitaNet-M / -L steps (GPU box):  bash tools/quick_stats_ml.sh TAG l 5
set -u
TAG=${1:-ml}; SIZE=${2:-l}; NB=${3:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- python tools/profile_ml.py $SIZE $NB ${4:-bf16} > gpurun_out/${TAG}.log 2>&1
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_trace
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
steps=6
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:26]:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{r['Name'][:100]:100s} {n/steps:6.1f}/step {t/n/1e3:9.1f}us {t/steps/1e6:8.3f} ms/step {t/tot*100:5.1f}%")
print("total ms/step", tot/steps/1e6)
PY
