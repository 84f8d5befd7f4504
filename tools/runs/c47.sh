cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_post.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/c47_test.log
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/c47_bench.json 2> gpurun_out/c47_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --steps 20 > gpurun_out/c47_bench20.json 2> gpurun_out/c47_bench20.err
