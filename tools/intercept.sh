#!/bin/bash
# fixed per-launch cost of every kernel of the headline step (GPU box): kernel stats at batch 128 and 256, intercept = 2 t(128) - t(256)
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for B in 128 256; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- \
    python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-ceiling --median-steps 0 > /dev/null 2> gpurun_out/${TAG}_trace.log
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_b${B}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_trace
done
python - <<PY
import csv
def load(p):
    rows = list(csv.DictReader(open(p)))
    steps = max([int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']] + [1])
    return {r['Name']: (int(r['Calls']) / steps, float(r['TotalDurationNs']) / int(r['Calls']) / 1e3) for r in rows}
a = load("gpurun_out/${TAG}_b128_kernel_stats.csv"); b = load("gpurun_out/${TAG}_b256_kernel_stats.csv")
out = []
for k, (n, t) in b.items():
    if k in a and abs(a[k][0] - n) < 0.01:
        ic = 2 * a[k][1] - t
        out.append((ic * n, k, n, a[k][1], t, ic))
out.sort(reverse=True)
print(f"{'kernel':80s} {'calls':>6s} {'t128':>8s} {'t256':>8s} {'fixed':>7s} {'fixed/step':>10s}")
for tot, k, n, ta, tb, ic in out[:30]:
    print(f"{k[:80]:80s} {n:6.1f} {ta:8.1f} {tb:8.1f} {ic:7.1f} {tot:10.1f}")
print("sum of fixed us/step", sum(o[0] for o in out), " step kernel-sum 256:", sum(n * t for n, t in b.values()))
PY
