cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "decoder or msda" 2>&1 | tail -8 > gpurun_out/c44_test.log
bash tools/pmc_all.sh > gpurun_out/c44_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/c44_prof.log 2>&1)
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python tools/top_dispatches.py $f 5 > gpurun_out/c44_top.txt 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/c44_kernel_stats.csv
