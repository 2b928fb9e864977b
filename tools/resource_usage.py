#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of the HIP sources (hipcc -Rpass-analysis=kernel-resource-usage),
as a table; `--check` fails if a hot-path kernel (name matches HOT) spills.

    python tools/resource_usage.py [--check] [--json profiles/rNN_resource_usage.json] [file.hip ...]
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "titanet_amd", "csrc")
sys.path.insert(0, ROOT)
from titanet_amd.csrc.build import FLAGS, SOURCES  # noqa: E402

HOT = re.compile(r"_v2|_v3|_v4|_v5|_v6|wide|dgrad|wgrad_batched|pgemm|slab")
# documented exceptions (spilled VGPRs allowed): everything else on the hot path must not spill
ALLOW = {}


def allowed(name):
    for k, v in ALLOW.items():
        if name.startswith(k):
            return v
    return 0


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def usage(src):
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c",
                                                                        os.path.join(CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            m = re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
            if not m:
                continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:"):
            cur = {"name": txt.split(":", 1)[1].strip(), "file": src}
            rows.append(cur)
        elif cur is not None and ":" in txt:
            k, v = txt.split(":", 1)
            try:
                cur[k.strip()] = int(v.strip())
            except ValueError:
                cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(.*", "", n).replace("void ", "")
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    check = "--check" in sys.argv
    jpath = None
    if "--json" in sys.argv:
        jpath = sys.argv[sys.argv.index("--json") + 1]
        args = [a for a in args if a != jpath]
    from concurrent.futures import ThreadPoolExecutor
    rows = []
    with ThreadPoolExecutor(max_workers=4) as ex:
        for r in ex.map(usage, [os.path.basename(a) for a in (args or SOURCES)]):
            rows += r
    rows.sort(key=lambda r: (-r.get("VGPRs Spill", 0), -r.get("VGPRs", 0)))
    print(f"{'kernel':72s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'scratch':>8s} {'SGPR':>5s} {'occ':>4s} {'LDS':>7s}")
    bad = []
    for r in rows:
        print(f"{r['name'][:72]:72s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('VGPRs Spill', 0):6d} "
              f"{r.get('ScratchSize [bytes/lane]', 0):8d} {r.get('TotalSGPRs', 0):5d} {r.get('Occupancy [waves/SIMD]', 0):4d} "
              f"{r.get('LDS Size [bytes/block]', 0):7d}")
        if HOT.search(r["name"]) and r.get("VGPRs Spill", 0) > allowed(r["name"]):
            bad.append(f"{r['name']}: {r.get('VGPRs Spill', 0)} spilled VGPRs (allowed {allowed(r['name'])})")
    if jpath:
        with open(jpath, "w") as fh:
            json.dump(rows, fh, indent=1)
    if check and bad:
        print("SPILLS in hot kernels:", *bad, sep="\n  ")
        sys.exit(1)


if __name__ == "__main__":
    main()
