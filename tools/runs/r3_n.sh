#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 500 python tools/dec_err_full.py vit32 bi32 2>&1 | grep -v amdgpu.ids | tail -15
timeout 500 python tools/dec_err_full.py vit32 2>&1 | grep -v amdgpu.ids | tail -3
