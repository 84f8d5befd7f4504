# round 3: PMC passes of the two dominant kernels, kernel-trace summary of the default bench, the bench line itself
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 bash tools/pmc_kernel.sh "python tools/bench_gemm_one.py qkv split" gemm_kernel:gemm_qkv_split > gpurun_out/p_pmc_gemm.log 2>&1
timeout 900 bash tools/pmc_kernel.sh "python tools/bench_gemm_one.py fc2 split" gemm_kernel:gemm_fc2_split > gpurun_out/p_pmc_gemm2.log 2>&1
timeout 900 bash tools/pmc_kernel.sh "python tools/bench_attn_split.py" vit_attn_split_kernel:attn_split > gpurun_out/p_pmc_attn.log 2>&1
python tools/pmc_summary.py gemm_qkv_split gemm_fc2_split attn_split > gpurun_out/r03_pmc_kernels.json 2> gpurun_out/p_pmc_summary.err
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/p_prof_bench.json 2> gpurun_out/p_prof.err)
cd $GRAFT_REPO_ROOT
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r03_bench_vith_bs8_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $GRAFT_REPO_ROOT/gpurun_out/p_prof_timed.json 2> $GRAFT_REPO_ROOT/gpurun_out/p_prof2.err)
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/r03_bench_vith_bs8_last5_forwards.txt 2>&1
