cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for d in 0 8; do for w in 4 8; do echo "defer $d waves $w"; HIPIE_FA_DEFER=$d HIPIE_FA_WAVES=$w DT=f16 timeout 200 python tools/bench_xattn.py 2>&1 | grep bi_xattn; done; done > gpurun_out/c50_xattn.log
