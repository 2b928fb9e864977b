#!/bin/bash
# SQ activity counters of the bench kernels (two --pmc passes; summaries under gpurun_out/, copied to profiles/ by hand)
set -u
TAG=${1:-r01_sq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/${TAG}_a -o a -- \
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --median-steps 0 > /dev/null 2> gpurun_out/${TAG}_a.log
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${TAG}_b -o b -- \
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --median-steps 0 > /dev/null 2> gpurun_out/${TAG}_b.log
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM --output-format csv -d gpurun_out/${TAG}_c -o c -- \
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --median-steps 0 > /dev/null 2> gpurun_out/${TAG}_c.log
python tools/pmc_summary.py gpurun_out/${TAG}_counters.json gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c > /dev/null
rm -rf gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_counters.json"))
keys = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
        "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM"]
rows = sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1].get("launches", 0))[:14]
for k, v in rows:
    print(k[:44].ljust(44), " ".join("%s=%.3g" % (c.replace("SQ_", ""), v.get(c, float("nan"))) for c in keys))
PY
