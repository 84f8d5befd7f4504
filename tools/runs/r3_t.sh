#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 500 python bench.py --steps 100 --warmup 2 --graph --no-cpu-baseline --no-parity-leg > gpurun_out/t_graph.json 2> gpurun_out/t_graph.err; echo "rc=$?"; tail -c 400 gpurun_out/t_graph.err
python - <<'PY'
import json
for l in open("gpurun_out/t_graph.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["config"]["launch"], d["steps"])
PY
