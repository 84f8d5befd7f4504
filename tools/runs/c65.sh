cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in 1 0 1 0; do echo "fused_heads $v: $(HIPIE_FUSED_HEADS=$v timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --timed-only 2>/dev/null)"; done > gpurun_out/c65_ab.log
