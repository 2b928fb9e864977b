#!/bin/bash
# kernel-time summary of the headline step (GPU box):  bash tools/prof_headline.sh TAG [grep pattern]
set -u
TAG=$1; PAT=${2:-.}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- \
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-ceiling --median-steps 0 > gpurun_out/${TAG}_bench_profiled.json 2> gpurun_out/${TAG}_trace.log
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_trace
python - <<PY
import csv, re
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
steps = max([int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']] + [1])
print("steps", steps, "total us/step", sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e3)
for r in rows:
    if not re.search(r"${PAT}", r['Name']): continue
    n = int(r['Calls']); t = float(r['TotalDurationNs'])
    print(f"{r['Name'][:90]:90s} {n / steps:6.1f} {t / n / 1e3:9.1f}us {t / steps / 1e3:9.1f}us/step")
PY
