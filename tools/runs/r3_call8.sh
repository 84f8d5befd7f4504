cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_e2e.py -q -m gpu -x -k "gemm or split or vit_backbone" 2>&1 | tail -5 > gpurun_out/c8_test.log
timeout 600 python tools/stage_times.py split3 > gpurun_out/c8_stage_split3.log 2>&1
timeout 400 python tools/deep_err.py split3 > gpurun_out/c8_deep_err.log 2>&1
