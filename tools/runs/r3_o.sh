#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "bi_i2t or vit_attention_split" 2>&1 | grep -E "bi_i2t|vit_attn_split|passed|failed|Error|error" | cut -c1-300
timeout 400 python tools/dec_err_full.py 2>&1 | grep -v amdgpu.ids | tail -14
timeout 300 python tools/bench_attn_split.py 2>&1 | tail -2
