# round 4, late: thin-K GEMM (gemm_k256.hip) -- correctness, micro A/B, the step and the full-size parity with it switched on
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 300 python -m pytest tests/test_gpu_gemm.py -q -x -k "thin_k256" 2>&1 | tail -4
timeout 200 python tools/bench_gemm_k256.py 2>&1 | tail -8
HIPIE_GEMM_K256=1 timeout 300 python bench.py --no-cpu-baseline --no-parity-leg 2>/dev/null | tail -1 | cut -c1-330
HIPIE_GEMM_K256=1 timeout 400 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "(full_size_split_policy and split3-e2e_full)" 2>&1 | grep -v "^$" | cut -c1-330 | tail -4
