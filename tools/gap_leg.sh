#!/bin/bash
# idle time between kernels of ONE other_configs leg (GPU box):  bash tools/gap_leg.sh TAG LEG
set -u
TAG=$1; LEG=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- \
    python -c "
import sys, json, torch
sys.path.insert(0, '.')
import bench
torch.set_num_threads(min(bench.effective_cores(), 8))
print(json.dumps(bench.other_configs(torch.device('cuda', 0), only=['$LEG'])))
" > gpurun_out/${TAG}_leg.json 2> gpurun_out/${TAG}_trace.log
F=$(find gpurun_out/${TAG}_trace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$F")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: the last 8 steps = from the 9th-last adam_kernel's end to the last adam_kernel's end
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = ad[-9] + 1, ad[-1]
seg = rows[lo:hi + 1]
t0 = int(rows[ad[-9]]["End_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
busy_end = t0; idle = 0; hist = collections.Counter(); after = collections.Counter()
prev = rows[ad[-9]]
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        g = s - busy_end; idle += g
        hist[min(int(g / 1000), 20)] += 1
        if g > 4000: after[prev["Kernel_Name"][:50] + " -> " + r["Kernel_Name"][:50]] += g
    busy_end = max(busy_end, e); prev = r
print("steps 8: wall per step us", (t1 - t0) / 8e3, " idle per step us", idle / 8e3, " kernels per step", len(seg) / 8)
print("gap histogram (us: count per step):", {k: round(v / 8, 1) for k, v in sorted(hist.items())})
for k, v in after.most_common(12): print(f"{v / 8e3:8.1f} us/step  {k}")
PY
rm -rf gpurun_out/${TAG}_trace
