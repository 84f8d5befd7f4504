#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_training.py -x -q -m gpu -k "train_step or split_linear_function" -s 2>&1 | grep -E "train step|passed|failed|Error|error" | tail -8 > gpurun_out/r6_train_step_test2.txt
timeout 1200 python tools/bench_train_step.py 2 3 2>&1 | grep "training step" > gpurun_out/r6_train_step_bench2.txt
