cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tests/study/debug_attn_split.py > gpurun_out/c3_dbg.log 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "split or hl8" 2>&1 | grep -E "vit_attn_split|passed|failed" > gpurun_out/c3_kernels.log
timeout 400 python tools/deep_err.py split3 > gpurun_out/c3_deep_err.log 2>&1
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu 2>&1 | tail -3 > gpurun_out/c3_gemm_test.log
