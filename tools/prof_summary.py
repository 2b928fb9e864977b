"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table
(calls, total ms, avg us, share).  Usage: python tools/prof_summary.py <results.db> [skip_first_n_steps_fraction]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values())
    print(f"{'kernel':110s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s}")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:110s} {n:6d} {t / 1e6:9.3f} {t / n / 1e3:9.2f} {100 * t / total:5.1f}%")
    print(f"{'TOTAL':110s} {sum(v[0] for v in agg.values()):6d} {total / 1e6:9.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
