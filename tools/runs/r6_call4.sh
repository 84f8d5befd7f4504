#!/bin/bash
# round 6: GEMM tile-order A/B (groups of g M-panels, M fastest inside a group) on the four ViT shapes, same box, alternating
mkdir -p gpurun_out
{
for g in 0 8 4 16 0 8; do echo "== HIPIE_GEMM_GROUP_M=$g"; HIPIE_GEMM_GROUP_M=$g timeout 300 python tools/ab_round5.py 2>/dev/null | grep -E "^gemm (qkv|fc1 +-> gelu)"; done
} > gpurun_out/r6_tile_order.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6_bench_line_b.json 2> gpurun_out/r6_bench_b.err
