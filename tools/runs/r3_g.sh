#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split or msda_backward_through" 2>&1 | tail -6
timeout 300 python tools/bench_attn_split.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "split or deep" 2>&1 | tail -4
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_g.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/bench_g.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["parity_err"]["max"], d["roofline_attention"]["avg_launch_ms"], d["roofline_attention"]["frac"])
PY
timeout 900 bash tools/pmc_kernel.sh "python tools/bench_attn_split.py global" vit_attn_split_kernel:attn_split > gpurun_out/g_pmc_attn.log 2>&1
python tools/pmc_summary.py attn_split > gpurun_out/g_pmc_attn.json 2> gpurun_out/g_pmc.err; cat gpurun_out/g_pmc_attn.json
