cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for hm in 0 1; do for sg in 1.5 4; do echo "head_major $hm sigma $sg"; HIPIE_MSDA_HEAD_MAJOR=$hm SIGMA=$sg timeout 200 python tools/bench_msda.py 2>&1 | grep msda_fused; done; done > gpurun_out/c60_msda.log
HIPIE_MSDA_HEAD_MAJOR=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "msda" 2>&1 | tail -3 >> gpurun_out/c60_msda.log
