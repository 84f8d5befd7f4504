#!/bin/bash
# round 6: the new tests (qkv geometry, literal configs[0] / full-size configs[1], MaskCLIP at full size) and the default bench line with the new legs
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/ -x -q -m gpu -s -k "r50_512 or full_size_r50 or follow_the_geometry or maskclip_full" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/r6_new_tests.txt
timeout 1500 python bench.py > gpurun_out/r6_bench_line_a.json 2> gpurun_out/r6_bench_a.err
tail -5 gpurun_out/r6_bench_a.err
