#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_post.py -x -q -s 2>&1 | grep -v "^$" | tail -8 > gpurun_out/r6_post_tests.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6_bench_line_c.json 2> gpurun_out/r6_bench_c.err
grep "leg failed" gpurun_out/r6_bench_c.err
