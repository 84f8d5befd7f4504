#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_post.py tests/test_gpu_gemm.py -x -q -s 2>&1 | grep -E "maskclip|passed|failed|Error" | tail -8 > gpurun_out/r6_post_tests2.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6_bench_line_d.json 2> gpurun_out/r6_bench_d.err
grep "leg failed" gpurun_out/r6_bench_d.err
