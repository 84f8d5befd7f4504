cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DT=f16 bash tools/pmc_kernel.sh "python tools/bench_xattn.py" xattn_i2t_kernel:xattn_i2t xattn_t2i_kernel:xattn_t2i > gpurun_out/c57_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python bench.py --no-cpu-baseline --no-parity-leg --timed-only > gpurun_out/c57_prof.log 2>&1)
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python tools/top_dispatches.py $f 5 > gpurun_out/c57_top.txt 2>&1
