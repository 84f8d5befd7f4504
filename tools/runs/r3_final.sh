#!/bin/bash
# round-3 closing run: whole GPU suite, smoke, the default bench line, kernel statistics of the timed loop, stage times
export HIPIE_MIOPEN_FIND=0
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) > gpurun_out/f_pytest.txt 2>&1; tail -16 gpurun_out/f_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/f_smoke.txt
unset HIPIE_MIOPEN_FIND
( time timeout 900 python bench.py > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
for l in open("gpurun_out/f_bench.json"):
    if l.startswith("{"):
        d=json.loads(l); print("BENCH", d["value"], d["ms_per_step"], "parity", d["parity_err"]["max"], {k:v["max"] for k,v in d["parity_err"]["fixtures"].items()}, "roofline", d["roofline"]["frac"], d["roofline_attention"]["frac"], "fast", (d.get("fast_policy") or {}).get("value")); print("CPU", json.dumps(d["cpu_baseline"])[:500])
PY
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b -- python $R/bench.py --no-cpu-baseline --no-parity-leg > $R/gpurun_out/f_prof_bench.json 2> $R/gpurun_out/f_prof.err)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/f_kernel_stats.csv 2>/dev/null
python tools/top_dispatches.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/f_last5_forwards.txt 2>&1; head -16 gpurun_out/f_last5_forwards.txt
timeout 300 python tools/stage_times.py split3 shapes > gpurun_out/f_stage.txt 2>&1; grep -E "forward_raw|backbone|DINO|BERT|bi_i2t|attn_f32|vit_attn" gpurun_out/f_stage.txt | head -12
