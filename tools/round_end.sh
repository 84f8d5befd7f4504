#!/bin/bash
# The measurement set behind profiles/rNN_* (run on the GPU box through gpurun; ROUND=06 names the files):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'ROUND=06 bash tools/round_end.sh'
# whole -m gpu suite + smoke, the default bench line (all legs), the other BASELINE configs, the kernel-trace summary of the default bench
# and of its timed steps, PMC passes of the two dominant kernels (separate --pmc passes, --kernel-trace only), stage times, the training step.
R=${ROUND:-06}
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/round_end; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -rs 2>&1 | tail -8 > $O/r${R}_gpu_suite_tail.txt; cat $O/r${R}_gpu_suite_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/r${R}_smoke.txt; cat $O/r${R}_smoke.txt
timeout 1200 python bench.py > $O/r${R}_bench_line.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --config 3 --steps 3 --warmup 2 --no-cpu-baseline --no-parity-leg > $O/r${R}_bench_config3_shard.json 2>/dev/null
timeout 400 python bench.py --config 4 --steps 3 --warmup 2 --no-cpu-baseline --no-parity-leg > $O/r${R}_bench_config4.json 2>/dev/null
timeout 200 python bench.py --config 1 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg > $O/r${R}_bench_config1.json 2>/dev/null
timeout 200 python bench.py --config 0 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-leg > $O/r${R}_bench_config0.json 2>/dev/null
timeout 600 python bench.py --graph --timed-only --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/r${R}_bench_line_hipgraph.json
for f in $O/r${R}_bench_config*.json; do tail -1 $f | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python bench.py --no-cpu-baseline --no-parity-leg --no-train-leg > $O/r${R}_bench_line_under_rocprof.json 2> $O/prof.err)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r${R}_bench_vith_bs8_kernel_stats.csv
(timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $O/prof_timed.json 2> $O/prof2.err)
cd $GRAFT_REPO_ROOT
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > $O/r${R}_bench_vith_bs8_last5_forwards.txt 2>&1
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_gemm_one.py qkv split" gemm_kernel:gemm_qkv_split > $O/pmc_gemm.log 2>&1
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_attn_split.py global" vit_attn_split_kernel:attn_split > $O/pmc_attn.log 2>&1
python tools/pmc_summary.py gemm_qkv_split attn_split > $O/r${R}_pmc_kernels.json 2> $O/pmc_summary.err
cat gpurun_out/pmc_gemm_qkv_split.txt gpurun_out/pmc_attn_split.txt > $O/r${R}_pmc_raw.txt
timeout 300 python tools/stage_times.py split3 shapes > $O/stage.log 2>&1; cp gpurun_out/stage_times.txt $O/r${R}_stage_times.txt
timeout 900 python tools/bench_train_step.py 2 3 2>&1 | grep "training step" > $O/r${R}_train_step_vith.txt
timeout 300 tools/ubench/pk_f32_hazard 2>&1 | sed 's/v_mov_b32 v10[0-9], %[0-9]//g' > $O/r${R}_pk_f32_hazard_forms.txt
head -12 $O/r${R}_bench_vith_bs8_last5_forwards.txt; tail -12 $O/r${R}_stage_times.txt; cat $O/r${R}_train_step_vith.txt
