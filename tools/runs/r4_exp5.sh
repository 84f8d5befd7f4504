# round 4, late: thin-K GEMM, second version (X rows through per-wave LDS-DMA slices)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 300 python -m pytest tests/test_gpu_gemm.py -q -x -k "thin_k256" 2>&1 | tail -4
timeout 200 python tools/bench_gemm_k256.py 2>&1 | tail -8
