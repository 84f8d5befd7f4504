cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "dynamic_mask" 2>&1 | tail -15 > gpurun_out/c38_test.log
timeout 300 python tools/bench_einsum.py 2>&1 | grep dynamic > gpurun_out/c38_dm.log
