set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for N in 256 512 1024; do (timeout 300 tools/pgemm_harness 76800 $N 512 2>&1 | grep "^NT GEMM\|variant") ; done > gpurun_out/r05_rwgemm_n_sweep.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/rw_fetch -o f -- tools/pgemm_harness 76800 512 512 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/rw_write -o w -- tools/pgemm_harness 76800 512 512 > /dev/null 2>&1
python - <<'PY' > gpurun_out/r05_rwgemm_pmc.txt
import csv, glob, collections
for d, c in (("gpurun_out/rw_fetch", "FETCH_SIZE"), ("gpurun_out/rw_write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            agg[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        if "gemm" in k: print(c, k, "launches", len(v), "avg KiB", sum(v) / len(v), "(x2 for FETCH on gfx950)")
PY
rm -rf gpurun_out/rw_fetch gpurun_out/rw_write
(timeout 900 python -m pytest tests/test_model_sizes_gpu.py tests/test_mask_gpu.py tests/test_config3_gpu.py tests/test_fp8_gpu.py tests/test_trained_parity_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r05_pytest6.txt
cat gpurun_out/r05_rwgemm_n_sweep.txt gpurun_out/r05_rwgemm_pmc.txt gpurun_out/r05_pytest6.txt
