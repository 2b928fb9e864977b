TN_ASP_FUSED=1 bash tools/prof_headline.sh r05_aspf1 "zzz" > /dev/null 2>&1
TN_ASP_FUSED=0 bash tools/prof_headline.sh r05_aspf0 "zzz" > /dev/null 2>&1
TN_ASP_FUSED=1 bash tools/prof_headline.sh r05_aspf1b "zzz" > /dev/null 2>&1
python - <<'PY'
import csv
def load(tag):
    rows = list(csv.DictReader(open(f"gpurun_out/{tag}_kernel_stats.csv")))
    steps = max([int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']] + [1])
    return {r['Name']: float(r['TotalDurationNs']) / steps / 1e3 for r in rows}
a, b, c = load("r05_aspf1"), load("r05_aspf0"), load("r05_aspf1b")
names = sorted(set(a) | set(b), key=lambda n: -abs(a.get(n, 0) - b.get(n, 0)))
print("total fused %.1f  stored %.1f  fused(again) %.1f us/step" % (sum(a.values()), sum(b.values()), sum(c.values())))
for n in names[:24]:
    print(f"{n[:86]:86s} fused {a.get(n, 0):8.1f} {c.get(n, 0):8.1f}  stored {b.get(n, 0):8.1f}  diff {a.get(n, 0) - b.get(n, 0):+7.1f}")
PY
