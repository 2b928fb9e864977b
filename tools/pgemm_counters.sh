#!/bin/bash
# SQ counters of pgemm_nt_kernel and its compiled-out variants (tools/pgemm_harness M N K: DBG 1 no MFMA, 2 no DMA, 8 no stores, ...)
#   bash tools/pgemm_counters.sh TAG [M N K]      -> gpurun_out/TAG_counters.json + a table on stdout
set -u
TAG=${1:-r06_pgemm}; M=${2:-76800}; N=${3:-1024}; K=${4:-1024}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { timeout 900 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/${TAG}_$1 -o $1 -- tools/pgemm_harness $M $N $K > gpurun_out/${TAG}_$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
run b "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"
run c "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM"
run d "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE"
python tools/pmc_summary.py gpurun_out/${TAG}_counters.json gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c gpurun_out/${TAG}_d > /dev/null
grep -E "pgemm_nt_kernel|dbg " gpurun_out/${TAG}_a.log | cut -c1-100
rm -rf gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c gpurun_out/${TAG}_d
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_counters.json"))
keys = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY",
        "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_VMEM"]
print("kernel".ljust(36), " ".join(k.replace("SQ_", "")[:14].rjust(14) for k in keys))
for k, v in sorted(d.items()):
    if "pgemm_nt_kernel" not in k: continue
    print(k[:36].ljust(36), " ".join(("%.4g" % v.get(c, float("nan"))).rjust(14) for c in keys), " launches", v.get("launches"))
PY
