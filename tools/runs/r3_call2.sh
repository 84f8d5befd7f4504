# round 3, GPU call 2: split attention + HL8 LayerNorm kernels, the split policy end to end (tiny, long prompt, full depth)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu 2>&1 | tail -5 > gpurun_out/c2_gemm_test.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "split or hl8" 2>&1 | tail -40 > gpurun_out/c2_kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -k "split" 2>&1 | tail -40 > gpurun_out/c2_e2e.log
timeout 400 python tools/deep_err.py split3 > gpurun_out/c2_deep_err.log 2>&1
timeout 300 python tools/bench_gemm2.py > gpurun_out/c2_gemm_bench.log 2>&1
