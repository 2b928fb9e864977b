set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_train_compare_gpu.py 2>&1 | tail -15) > gpurun_out/r05_pytest1.txt
(timeout 300 tools/flush_probe 2>&1) > gpurun_out/r05_flush_probe.txt
(timeout 600 bash tools/ab_env.sh TN_OVERLAP 0 1 3 2>&1) > gpurun_out/r05_ab_overlap.txt
(timeout 900 python tools/train_compare.py 0.02 0.008 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_train_compare_sweep.txt
tail -5 gpurun_out/r05_pytest1.txt; cat gpurun_out/r05_flush_probe.txt gpurun_out/r05_ab_overlap.txt gpurun_out/r05_train_compare_sweep.txt
