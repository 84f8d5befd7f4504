#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python tools/bench_ffn_chunks.py 2>&1 | grep -v amdgpu | tee gpurun_out/s_ffn_chunks.txt
