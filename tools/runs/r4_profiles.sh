# round 4: PMC passes (split GEMM, fused FFN vs the two GEMMs it replaces), kernel-trace summaries of the default bench, stage times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_gemm_one.py qkv split" gemm_kernel:gemm_qkv_split > gpurun_out/p_pmc_gemm.log 2>&1
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_ffn_fused.py fused" ffn_fused_kernel:ffn_fused > gpurun_out/p_pmc_ffn.log 2>&1
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_ffn_fused.py two" gemm_kernel:ffn_two_gemms > gpurun_out/p_pmc_ffn2.log 2>&1
python tools/pmc_summary.py gemm_qkv_split ffn_fused ffn_two_gemms > gpurun_out/r04_pmc_kernels.json 2> gpurun_out/p_pmc_summary.err
cat gpurun_out/pmc_gemm_qkv_split.txt gpurun_out/pmc_ffn_fused.txt gpurun_out/pmc_ffn_two_gemms.txt > gpurun_out/r04_pmc_raw.txt
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_line_under_rocprof.json 2> gpurun_out/p_prof.err)
cd $GRAFT_REPO_ROOT
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r04_bench_vith_bs8_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $GRAFT_REPO_ROOT/gpurun_out/p_prof_timed.json 2> $GRAFT_REPO_ROOT/gpurun_out/p_prof2.err)
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/r04_bench_vith_bs8_last5_forwards.txt 2>&1
timeout 300 python tools/stage_times.py split3 shapes > gpurun_out/r04_stage_times.txt 2>&1
head -30 gpurun_out/r04_pmc_kernels.json; head -12 gpurun_out/r04_bench_vith_bs8_kernel_stats.csv | cut -c1-150; tail -c 400 gpurun_out/r04_bench_line_under_rocprof.json
