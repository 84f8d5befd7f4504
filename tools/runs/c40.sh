cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/c40_test.log
for v in 0 1 2 3; do echo "variant $v"; HIPIE_DM_VARIANT=$v timeout 300 python tools/bench_einsum.py 2>&1 | grep dynamic_mask16; done > gpurun_out/c40_dm.log
