# round 4, final: the default bench line (all legs), configs 3 / 4 / 1 / 0 lines, the whole GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 1500 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; echo "bench rc=$?"
timeout 400 python bench.py --config 3 --steps 3 --warmup 2 --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_config3_shard.json 2>/dev/null
timeout 600 python bench.py --config 4 --steps 3 --warmup 2 --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_config4.json 2>/dev/null
timeout 300 python bench.py --config 1 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_config1.json 2>/dev/null
timeout 300 python bench.py --config 0 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_config0.json 2>/dev/null
for f in gpurun_out/r04_bench_config*.json; do tail -1 $f | cut -c1-200; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_gpu_suite_tail.txt; cat gpurun_out/r04_gpu_suite_tail.txt
tail -1 gpurun_out/r04_bench_line.json | cut -c1-400
