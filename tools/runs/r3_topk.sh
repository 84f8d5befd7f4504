#!/bin/bash
# top-k kernel: unit tests, e2e tests that free-run the selection, then 200-step graph replays of both policies
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "topk" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_post.py -q -m gpu -x 2>&1 | tail -5
timeout 600 python bench.py --steps 200 --warmup 3 --graph --no-cpu-baseline > gpurun_out/graph_split.log 2>&1; echo "graph split rc=$?"; tail -c 1500 gpurun_out/graph_split.log
timeout 600 python bench.py --steps 200 --warmup 3 --graph --precision fast --no-cpu-baseline > gpurun_out/graph_fast.log 2>&1; echo "graph fast rc=$?"; tail -c 600 gpurun_out/graph_fast.log
