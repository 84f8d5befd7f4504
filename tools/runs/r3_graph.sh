cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python bench.py --graph --steps 200 --warmup 2 --no-cpu-baseline --no-parity-leg --timed-only > gpurun_out/g_split.json 2> gpurun_out/g_split.err; echo "rc=$?" >> gpurun_out/g_split.err
timeout 600 python bench.py --steps 50 --warmup 2 --no-cpu-baseline --no-parity-leg --timed-only > gpurun_out/g_split_eager.json 2> gpurun_out/g_split_eager.err
timeout 600 python bench.py --precision fast --graph --steps 200 --warmup 2 --no-cpu-baseline --no-parity-leg --timed-only > gpurun_out/g_fast.json 2> gpurun_out/g_fast.err; echo "rc=$?" >> gpurun_out/g_fast.err
timeout 600 python bench.py --precision fast --steps 50 --warmup 2 --no-cpu-baseline --no-parity-leg --timed-only > gpurun_out/g_fast_eager.json 2> gpurun_out/g_fast_eager.err
