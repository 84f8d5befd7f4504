#!/bin/bash
# kernel trace of the timed loop only: dispatch census after the top-k / fp32-A GEMM changes
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o b -- python $R/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $R/gpurun_out/d_prof_timed.json 2> $R/gpurun_out/d_prof2.err)
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/d_last5_forwards.txt 2>&1
head -5 gpurun_out/d_last5_forwards.txt
