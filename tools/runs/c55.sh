cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for sp in 4 6 8 12 16; do echo "SP $sp"; HIPIE_XT_SP=$sp DT=f16 timeout 200 python tools/bench_xattn.py 2>&1 | grep bi_xattn; done > gpurun_out/c55_xattn.log
