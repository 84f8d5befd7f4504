cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm.py -q -m gpu -s -k "split or hl8 or gemm" 2>&1 | grep -E "vit_attn_split|passed|failed|FAILED|Error" > gpurun_out/c4_kernels.log
timeout 600 python tools/stage_times.py split3 > gpurun_out/c4_stage_split3.log 2>&1
timeout 400 python tools/deep_err.py split3 > gpurun_out/c4_deep_err.log 2>&1
