#!/bin/bash
export HIPIE_MIOPEN_FIND=0
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "topk" 2>&1 | tail -2
timeout 600 python tools/deep_err.py split3 parity fast fixtures=e2e_full,e2e_deep 2>&1 | grep -v amdgpu > gpurun_out/p_policy_errors.txt; cat gpurun_out/p_policy_errors.txt
unset HIPIE_MIOPEN_FIND
timeout 300 python bench.py --steps 3 --warmup 1 --classes 150 --text-len 815 --no-cpu-baseline --no-parity-leg > gpurun_out/p_cfg3.json 2>gpurun_out/p_cfg3.err; tail -c 300 gpurun_out/p_cfg3.err
timeout 400 python bench.py --steps 3 --warmup 1 --classes 1203 --text-len 4096 --size 1344 --no-cpu-baseline --no-parity-leg > gpurun_out/p_cfg4.json 2>gpurun_out/p_cfg4.err; tail -c 300 gpurun_out/p_cfg4.err
python - <<'PY'
import json
for f in ("p_cfg3","p_cfg4"):
    for l in open("gpurun_out/%s.json"%f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["value"], d["ms_per_step"])
PY
