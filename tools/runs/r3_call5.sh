cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/bench_gemm2.py 32768 variants > gpurun_out/c5_variants.log 2>&1
timeout 300 python tools/bench_gemm2.py > gpurun_out/c5_gemm_bench.log 2>&1
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu 2>&1 | tail -3 > gpurun_out/c5_gemm_test.log
