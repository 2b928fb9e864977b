#!/bin/bash
# SQ / TCC activity counters of an arbitrary command (GPU box): separate --pmc passes, per-kernel summary.
#   usage: bash tools/sq_counters.sh TAG cmd args...       -> gpurun_out/TAG_counters.json
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
pass() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${TAG}_$name -o $name -- "${CMD[@]}" > /dev/null 2> gpurun_out/${TAG}_$name.log
}
CMD=("$@")
pass a SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS
pass b SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES
pass c SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM
pass d SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass e FETCH_SIZE
pass f WRITE_SIZE
pass g TCC_HIT_sum TCC_MISS_sum
python tools/pmc_summary.py gpurun_out/${TAG}_counters.json gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c gpurun_out/${TAG}_d gpurun_out/${TAG}_e gpurun_out/${TAG}_f gpurun_out/${TAG}_g > /dev/null
rm -rf gpurun_out/${TAG}_[a-g]
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_counters.json"))
d.pop("_meta", None)
rows = sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1].get("launches", 0))[:12]
for k, v in rows:
    print(k[:60])
    print("   ", " ".join("%s=%.4g" % (c.replace("SQ_", ""), x) for c, x in sorted(v.items())))
PY
