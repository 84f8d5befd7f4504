#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 400 python tools/dec_err_full.py bi32 2>&1 | grep -v amdgpu.ids | tail -15
