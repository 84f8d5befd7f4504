# round 4, late: the LDS-DMA form of the fp32-feature mask contraction (hipie_mask_einsum_ws), pyramid levels kept as views
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k "mask_einsum" 2>&1 | tail -4
timeout 150 python tools/bench_einsum.py 2>&1 | head -4
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "(full_size_split_policy and split3-e2e_full) or (stages and split3) or r50_tiny" 2>&1 | grep -v "^$" | cut -c1-400 | tail -12
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg 2>/dev/null | tail -1 | cut -c1-420
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $GRAFT_REPO_ROOT/gpurun_out/x2_prof_timed.json 2> $GRAFT_REPO_ROOT/gpurun_out/x2_prof.err)
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/x2_last5_forwards.txt 2>&1
head -3 gpurun_out/x2_last5_forwards.txt; grep -n "mask_einsum\|msda_d32\|me_split" gpurun_out/x2_last5_forwards.txt | head -5 | cut -c1-150
sed -n '/elementwise \/ copy kernels by launch size/,$p' gpurun_out/x2_last5_forwards.txt | cut -c1-120 | head -14
