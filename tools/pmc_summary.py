"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per counter pass) per kernel.
Usage: python tools/pmc_summary.py out.json DIR1 DIR2 ..."""
import collections, csv, glob, json, sys

out = {}
for d in sys.argv[2:]:
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
json.dump(out, open(sys.argv[1], "w"), indent=1)
top = sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0) * kv[1]["launches"])[:16]
for k, v in top:
    print("%-70s n=%5d FETCH_SIZE=%12.1f WRITE_SIZE=%12.1f" % (k, v["launches"], v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
