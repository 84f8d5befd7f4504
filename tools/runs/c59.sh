cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -x -q -m gpu -k "not full_size" 2>&1 | tail -6 > gpurun_out/c59_test.log
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/c59_bench.json 2> gpurun_out/c59_bench.err
