#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "msda" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "split or deep or stages" 2>&1 | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_f.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/bench_f.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["parity_err"]["max"], d["parity_err"]["fixtures"]["e2e_deep"]["per_output"])
PY
