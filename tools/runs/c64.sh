cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q -m gpu -k "not full_size" 2>&1 | tail -8 > gpurun_out/c64_test.log
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/c64_bench.json 2> gpurun_out/c64_bench.err
