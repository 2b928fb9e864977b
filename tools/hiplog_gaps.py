"""largest time jumps between consecutive lines of an AMD_LOG_LEVEL log (':3:file:line: <us> us: ...')"""
import re, sys
prev = None; out = []
pat = re.compile(r":\s*(\d+)\s*us:")
lines = open(sys.argv[1], errors="replace").read().splitlines()
for i, ln in enumerate(lines):
    m = pat.search(ln)
    if not m: continue
    t = int(m.group(1))
    if prev is not None and t - prev[0] > 20000:
        out.append((t - prev[0], prev[1], i))
    prev = (t, i)
for d, a, b in sorted(out, reverse=True)[:8]:
    print(f"--- {d/1e3:.1f} ms between log lines {a} and {b}")
    for ln in lines[max(0, a - 3):b + 2]: print("   ", ln[:220])
