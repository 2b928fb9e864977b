set -x
mkdir -p gpurun_out
T=tests/test_model_sizes_gpu.py::test_wide_model_gradients_at_scale_agree_with_generic_path
(timeout 300 tools/pgemm_harness 76800 512 512 2>&1 | grep "variant") > gpurun_out/r05_dbg_a.txt
(timeout 300 tools/pgemm_harness 64000 512 512 2>&1 | grep "variant") >> gpurun_out/r05_dbg_a.txt
(timeout 300 python -m pytest "$T" -x -q -s 2>&1 | grep -v amdgpu | tail -15) > gpurun_out/r05_dbg_b.txt
(TN_RW_VARIANT=1 timeout 300 python -m pytest "$T" -x -q -s 2>&1 | grep -v amdgpu | tail -8) > gpurun_out/r05_dbg_c.txt
(TN_KROT=0 timeout 300 python -m pytest "$T" -x -q -s 2>&1 | grep -v amdgpu | tail -8) > gpurun_out/r05_dbg_d.txt
(timeout 300 tools/dgrad_dw_harness 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl") > gpurun_out/r05_dgrad_dw_md.txt
(timeout 600 python -m pytest tests/test_v2_shapes_gpu.py tests/test_forward_gpu.py tests/test_backward_gpu.py tests/test_bench_shape_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r05_pytest5.txt
(timeout 600 bash tools/ab_bench.sh lib_v11 2 2>&1) > gpurun_out/r05_ab_s1.txt
cat gpurun_out/r05_dbg_a.txt gpurun_out/r05_dbg_b.txt gpurun_out/r05_dbg_c.txt gpurun_out/r05_dbg_d.txt gpurun_out/r05_dgrad_dw_md.txt gpurun_out/r05_pytest5.txt gpurun_out/r05_ab_s1.txt
