cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for a in 0 1 2 4 8 16 31; do echo "abl $a"; HIPIE_XA_ABL=$a DT=f16 timeout 200 python tools/bench_xattn.py 2>&1 | grep bi_xattn; done > gpurun_out/c52_xattn.log
