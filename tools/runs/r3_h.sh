#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "pointwise" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "split or deep or stages" 2>&1 | tail -4
for fh in 0 1; do
  HIPIE_FUSED_HEADS=$fh timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_h$fh.log 2>&1; echo "bench fused_heads=$fh rc=$?"
  python - <<PY
import json
for l in open("gpurun_out/bench_h$fh.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["parity_err"]["max"])
PY
done
