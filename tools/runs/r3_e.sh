#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/bench_gemm2.py 32768 epilogue > gpurun_out/e_epi.txt 2>&1; cat gpurun_out/e_epi.txt
