cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/c7_test.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/c7_smoke.log 2>&1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/c7_bench.json 2> gpurun_out/c7_bench.err
timeout 600 python tools/stage_times.py split3 > gpurun_out/c7_stage_split3.log 2>&1
