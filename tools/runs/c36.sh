cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/c36_test.log
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/c36_bench.json 2> gpurun_out/c36_bench.err
STACKS=1 timeout 300 python tools/torch_profile.py > gpurun_out/c36_prof.log 2>&1
