#!/bin/bash
# round 5, final tree (after the token-id staging, the v_exp_f32 settle and the tail on every attention instance): whole GPU suite, smoke,
# the default bench line (all legs), configs 0 / 4 (the two the late changes touch), the profile set again
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5f2
O=gpurun_out/r5f2
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/r05_gpu_suite_tail.txt; cat $O/r05_gpu_suite_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/r05_bench_line.json 2> $O/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --config 4 --steps 3 --warmup 2 --no-cpu-baseline --no-parity-leg > $O/r05_bench_config4.json 2>/dev/null
timeout 200 python bench.py --config 0 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-leg > $O/r05_bench_config0.json 2>/dev/null
for f in $O/r05_bench_config*.json; do tail -1 $f | cut -c1-220; done
tail -1 $O/r05_bench_line.json | cut -c1-300
bash tools/runs/r5_profiles.sh > $O/profiles.log 2>&1; tail -3 $O/profiles.log
