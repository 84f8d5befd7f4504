cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "xattn" 2>&1 | tail -12 > gpurun_out/c54_test.log
DT=f16 timeout 200 python tools/bench_xattn.py 2>&1 | grep bi_xattn > gpurun_out/c54_xattn.log
