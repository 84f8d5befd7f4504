#!/bin/bash
# round 6: kernel trace of the timed steps of the default bench workload (two streams), top dispatch table of the last 5 forwards
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
O=$GRAFT_REPO_ROOT/gpurun_out/r6p
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $O/prof_timed.json 2> $O/prof2.err
cd $GRAFT_REPO_ROOT
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $O/r06_bench_vith_bs8_timed_kernel_stats.csv
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > $O/r06_bench_vith_bs8_last5_forwards.txt 2>&1
head -60 $O/r06_bench_vith_bs8_last5_forwards.txt
