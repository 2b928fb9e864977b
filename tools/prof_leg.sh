#!/bin/bash
# kernel-time summary of ONE other_configs leg of bench.py (GPU box):  bash tools/prof_leg.sh TAG LEG
set -u
TAG=$1; LEG=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- \
    python -c "
import sys, json, torch
sys.path.insert(0, '.')
import bench
torch.set_num_threads(min(bench.effective_cores(), 8))
print(json.dumps(bench.other_configs(torch.device('cuda', 0), only=['$LEG'])))
" > gpurun_out/${TAG}_leg.json 2> gpurun_out/${TAG}_trace.log
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_trace
tail -1 gpurun_out/${TAG}_leg.json
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
steps = max([int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']] + [1])
print("steps", steps)
for r in rows[:36]:
    n = int(r['Calls']); t = float(r['TotalDurationNs'])
    print(f"{r['Name'][:100]:100s} {n / steps:6.1f} {t / n / 1e3:9.1f}us {t / steps / 1e3:9.1f}us/step")
print("total us/step", sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e3)
PY
