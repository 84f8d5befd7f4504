#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 300 python tools/conv_precision.py > gpurun_out/conv_precision.txt 2>&1; cat gpurun_out/conv_precision.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -k "stages_on_the_gpu and full" 2>&1 | grep -E "stages|passed|failed|Error" | cut -c1-900
