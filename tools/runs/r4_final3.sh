# round 4, last call: the whole GPU suite and a short bench line of the final tree (thin-K GEMM on for N >= 384, training host logic added)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_gpu_suite_tail.txt; cat gpurun_out/r04_gpu_suite_tail.txt
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/r04_bench_line_final_tree.json 2>/dev/null; tail -1 gpurun_out/r04_bench_line_final_tree.json | cut -c1-330
