# quick check after a decoder-side kernel change: decoder-relevant tests + fuzz subset + headline profile
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x -k "parity or masked or shapes or asp_fused" 2>&1 | grep -v amdgpu | tail -4)
(timeout 600 python tools/fuzz_paths.py 8 5 2>&1 | grep -E "^(ok|FAIL|worst)" | cut -c1-40,250-400)
bash tools/prof_headline.sh ${1:-r05_q} "asp|wide|gemm_nt_kernel<unsigned short, 2, 4, ProdDy" 2>&1 | tail -12
