# the other BASELINE configurations, stage times and the kernel-trace summary of the default bench (end of round 2)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --classes 150 --text-len 815 --size 1344 > gpurun_out/f2_cfg3.json 2> gpurun_out/f2_cfg3.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --classes 1203 --text-len 4096 --size 1344 > gpurun_out/f2_cfg4.json 2> gpurun_out/f2_cfg4.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --model r50 --batch 4 > gpurun_out/f2_cfg1.json 2> gpurun_out/f2_cfg1.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --text-len 4096 --timed-only > gpurun_out/f2_pad4096.json 2> gpurun_out/f2_pad4096.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --task grounding --timed-only > gpurun_out/f2_grounding.json 2> gpurun_out/f2_grounding.err
timeout 300 python tools/stage_times.py > gpurun_out/f2_stage.log 2>&1
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/f2_prof.log 2>&1)
cd $GRAFT_REPO_ROOT
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/f2_kernel_stats.csv
