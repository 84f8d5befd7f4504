#!/bin/bash
# top-k unit tests, 84x84 split attention, fp32-A GEMM, then the e2e suites and a bench line
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm.py -q -m gpu -k "topk or split or fp32_rows" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x 2>&1 | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_b.log 2>&1; echo "bench rc=$?"; tail -c 900 gpurun_out/bench_b.log
timeout 300 python tools/stage_times.py split > gpurun_out/stage_b.txt 2>&1; head -24 gpurun_out/stage_b.txt
