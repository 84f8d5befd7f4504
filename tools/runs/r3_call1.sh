# round 3, GPU call 1: hipie_gemm correctness + rates, policy errors at full depth
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu -s 2>&1 | tail -80 > gpurun_out/c1_gemm_test.log
timeout 400 python tools/bench_gemm2.py > gpurun_out/c1_gemm_bench.log 2>&1
timeout 400 python tools/deep_err.py parity fast > gpurun_out/c1_deep_err.log 2>&1
