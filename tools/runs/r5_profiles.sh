#!/bin/bash
# round 5: kernel-trace summary of the default bench (whole process) + the line that run printed, the timed steps only (last 5 forwards),
# PMC passes of the two dominant kernels (split GEMM qkv shape, global split attention with the 16-row tail), stage times.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5p
O=gpurun_out/r5p
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python bench.py --no-cpu-baseline --no-parity-leg > $O/r05_bench_line_under_rocprof.json 2> $O/prof.err)
cd $GRAFT_REPO_ROOT
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r05_bench_vith_bs8_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $GRAFT_REPO_ROOT/$O/prof_timed.json 2> $GRAFT_REPO_ROOT/$O/prof2.err)
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > $O/r05_bench_vith_bs8_last5_forwards.txt 2>&1
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_gemm_one.py qkv split" gemm_kernel:gemm_qkv_split > $O/pmc_gemm.log 2>&1
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_attn_split.py global" vit_attn_split_kernel:attn_split > $O/pmc_attn.log 2>&1
python tools/pmc_summary.py gemm_qkv_split attn_split > $O/r05_pmc_kernels.json 2> $O/pmc_summary.err
cat gpurun_out/pmc_gemm_qkv_split.txt gpurun_out/pmc_attn_split.txt > $O/r05_pmc_raw.txt
timeout 300 python tools/stage_times.py split3 shapes > $O/stage.log 2>&1; cp gpurun_out/stage_times.txt $O/r05_stage_times.txt
head -14 $O/r05_bench_vith_bs8_last5_forwards.txt; cat $O/r05_pmc_kernels.json | head -60; tail -14 $O/r05_stage_times.txt
