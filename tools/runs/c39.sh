cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_selftest.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/c39_test.log
timeout 300 python tools/bench_einsum.py 2>&1 | grep dynamic > gpurun_out/c39_dm.log
DT=f16 timeout 300 python tools/bench_xattn.py > gpurun_out/c39_xattn.log 2>&1
