# round 4, late: mask contraction with the unpadded / swizzled image, three LDS tiles, XCD-aware workgroup map (A/B)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k "mask_einsum" 2>&1 | tail -3
timeout 150 python tools/bench_einsum.py 2>&1 | head -4
HIPIE_ME_XMAP=0 timeout 150 python tools/bench_einsum.py 2>&1 | head -1
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "(full_size_split_policy and split3-e2e_full)" 2>&1 | grep -v "^$" | cut -c1-400 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg 2>/dev/null | tail -1 | cut -c1-420
