#!/bin/bash
# checkpoint: the whole GPU suite, smoke, the default bench line
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) 2>&1 | tail -14
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1500 python bench.py > gpurun_out/bench_i.log 2> gpurun_out/bench_i.err ) 2>&1 | tail -4; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/bench_i.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], json.dumps(d["parity_err"])[:900]); print(json.dumps(d["cpu_baseline"])[:400]); print(d.get("fast_policy",{}).get("value"))
PY
