set -x
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | grep "passed\|failed\|FAILED\|rror" | tail -6) > gpurun_out/r05_pytest30.txt
cat gpurun_out/r05_pytest30.txt
(timeout 900 python tools/fuzz_paths.py 24 41 2>&1 | grep -E "^(ok|FAIL|worst)" | cut -c1-60,250-420) > gpurun_out/r05_fuzz30.txt
grep -c "^ok" gpurun_out/r05_fuzz30.txt; grep "^FAIL\|^worst" gpurun_out/r05_fuzz30.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
