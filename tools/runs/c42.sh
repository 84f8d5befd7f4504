cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg > gpurun_out/c42_eager.json 2> gpurun_out/c42_eager.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --graph > gpurun_out/c42_graph.json 2> gpurun_out/c42_graph.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --graph --steps 20 > gpurun_out/c42_graph20.json 2> gpurun_out/c42_graph20.err
