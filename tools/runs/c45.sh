cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "group_norm" 2>&1 | tail -8 > gpurun_out/c45_test.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -s -k "not full_size" 2>&1 | grep -v "^$" | tail -16 >> gpurun_out/c45_test.log
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/c45_bench.json 2> gpurun_out/c45_bench.err
