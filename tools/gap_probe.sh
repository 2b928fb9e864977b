#!/bin/bash
# kernel timeline gaps of a command (GPU box): bash tools/gap_probe.sh TAG cmd...   -> largest idle gaps between consecutive kernels
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- "$@" > gpurun_out/${TAG}.log 2> gpurun_out/${TAG}_trace.log
F=$(find gpurun_out/${TAG}_trace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
gaps=[]
for a,b in zip(rows,rows[1:]):
    g=int(b["Start_Timestamp"])-int(a["End_Timestamp"])
    gaps.append((g,a["Kernel_Name"][:60],b["Kernel_Name"][:60],int(a["End_Timestamp"])))
t0=int(rows[0]["Start_Timestamp"])
big=sorted(gaps,reverse=True)[:14]
for g,a,b,t in big: print(f"gap {g/1e6:8.2f} ms at t={(t-t0)/1e6:9.1f} ms  after [{a}]  before [{b}]")
print("kernels", len(rows), "total span ms", (int(rows[-1]["End_Timestamp"])-t0)/1e6, "sum of gaps > 0.1 ms:", sum(g for g,_,_,_ in gaps if g>1e5)/1e6)
PY
rm -rf gpurun_out/${TAG}_trace
