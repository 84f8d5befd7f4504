#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
O=$GRAFT_REPO_ROOT/gpurun_out/r6p
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o t -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py 2 3 > $O/train_prof.txt 2> $O/train_prof.err
cd $GRAFT_REPO_ROOT
cp $(find /tmp/kt3 -name "*kernel_stats.csv" | head -1) $O/r06_train_step_kernel_stats.csv
head -30 $O/r06_train_step_kernel_stats.csv | cut -c1-200
