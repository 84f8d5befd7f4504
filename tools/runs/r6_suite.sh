#!/bin/bash
# whole -m gpu suite + smoke on the current tree
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu -rs 2>&1 | tail -15 > gpurun_out/r6_gpu_suite_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/r6_smoke.txt
