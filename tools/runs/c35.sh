cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_post.py -x -q -m gpu -k "full_size or sem_pan or bf16" 2>&1 | tail -15 > gpurun_out/c35_test.log
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --classes 150 --text-len 815 --size 1344 > gpurun_out/c35_cfg3.json 2> gpurun_out/c35_cfg3.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --classes 1203 --text-len 4096 --size 1344 > gpurun_out/c35_cfg4.json 2> gpurun_out/c35_cfg4.err
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg --model r50 --batch 4 > gpurun_out/c35_cfg1.json 2> gpurun_out/c35_cfg1.err
