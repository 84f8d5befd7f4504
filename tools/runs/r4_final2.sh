# round 4, final (second edition, after the MSDA map / mask-contraction / pyramid-view commits): default bench line (all legs), kernel-trace
# summaries of the same build, configs 3 / 4 / 1 / 0 lines, the whole GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 1500 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; echo "bench rc=$?"
tail -1 gpurun_out/r04_bench_line.json | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_line_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/p_prof.err)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r04_bench_vith_bs8_kernel_stats.csv
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $GRAFT_REPO_ROOT/gpurun_out/p_prof_timed.json 2> $GRAFT_REPO_ROOT/gpurun_out/p_prof2.err)
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/r04_bench_vith_bs8_last5_forwards.txt 2>&1
head -8 gpurun_out/r04_bench_vith_bs8_kernel_stats.csv | cut -c1-150
timeout 400 python bench.py --config 3 --steps 3 --warmup 2 --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_config3_shard.json 2>/dev/null
timeout 600 python bench.py --config 4 --steps 3 --warmup 2 --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_config4.json 2>/dev/null
timeout 300 python bench.py --config 1 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_config1.json 2>/dev/null
timeout 300 python bench.py --config 0 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_config0.json 2>/dev/null
for f in gpurun_out/r04_bench_config*.json; do tail -1 $f | cut -c1-200; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_gpu_suite_tail.txt; cat gpurun_out/r04_gpu_suite_tail.txt
