#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attn_f32" 2>&1 | tail -4
timeout 400 python tools/dec_err_full.py 2>&1 | grep -v amdgpu.ids | tail -15
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -k "tiny_split or deep or bert or full_size_split or long_prompt" 2>&1 | grep -E "policy|passed|failed|Error|rror" | cut -c1-400
timeout 300 python tools/stage_times.py split3 2>&1 | grep -E "forward_raw|attn_f32|flash_attn|text_encoder|DINO decoder|MaskDINO decoder" 
