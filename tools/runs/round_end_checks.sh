# what was run on the GPU box at the end of round 2 (through gpurun): full GPU suite, smoke, the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/final_test.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
