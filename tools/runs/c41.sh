cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/stage_times.py > gpurun_out/c41_stage.log 2>&1
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python bench.py --no-cpu-baseline --no-parity-leg --timed-only > gpurun_out/c41_prof.log 2>&1)
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python tools/top_dispatches.py $f 5 > gpurun_out/c41_top.txt 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/c41_kernel_stats.csv
