"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per counter pass) per kernel.
Usage: python tools/pmc_summary.py out.json [--steps N] DIR1 DIR2 ...
--steps: training steps the profiled command ran (bench.py: warmup + 4 class-probe steps + steps), stored as _meta.steps so
that bench.py can turn per-launch averages x launches into HBM bytes per step."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

argv = sys.argv[1:]
steps = 0
if "--steps" in argv:
    i = argv.index("--steps")
    steps = int(argv[i + 1])
    del argv[i:i + 2]
out = {}
for d in argv[1:]:
    for path in glob.glob(d + "/*counter_collection.csv"):
        rows = list(csv.DictReader(open(path)))
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
            c = r["Counter_Name"]
            agg[(k, c)] += float(r["Counter_Value"]); n[(k, c)] += 1
        for (k, c), v in agg.items():
            out.setdefault(k, {})[c] = v / n[(k, c)]
            out[k]["launches"] = n[(k, c)]
total = sum((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 * v["launches"] for v in out.values())
try:
    from titanet_amd.csrc.build import _digest
    kd = _digest()[:16]
except Exception:
    kd = None
out["_meta"] = {"steps": steps, "unit": "KiB per launch (FETCH_SIZE to be doubled on gfx950)",
                "bytes_per_step": total / steps if steps else None,
                "kernel_digest": kd}      # bench.py prints this file's traffic only for the kernels it was taken with
json.dump(out, open(argv[0], "w"), indent=1)
if steps:
    print("HBM bytes per step (2*FETCH + WRITE): %.2f GB" % (total / steps / 1e9))
del out["_meta"]
top = sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0) * kv[1]["launches"])[:16]
for k, v in top:
    print("%-70s n=%5d FETCH_SIZE=%12.1f WRITE_SIZE=%12.1f" % (k, v["launches"], v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
