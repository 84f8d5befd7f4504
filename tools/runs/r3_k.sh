#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
for f in "" "mha32" "mha32 qkv32"; do
  echo "=== $f"; timeout 400 python tools/dec_err_full.py $f 2>&1 | grep -v amdgpu.ids | tail -16
done > gpurun_out/dec_err.txt 2>&1
cat gpurun_out/dec_err.txt
