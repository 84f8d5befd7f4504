#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 bash tools/pmc_kernel.sh "python tools/bench_attn_split.py global" vit_attn_split_kernel:attn_split > gpurun_out/q_pmc_attn.log 2>&1
python tools/pmc_summary.py attn_split > gpurun_out/q_pmc_attn.json 2> gpurun_out/q_pmc.err; cat gpurun_out/q_pmc_attn.json
timeout 200 python tools/bench_msda.py bwd 2>&1 | tail -4
