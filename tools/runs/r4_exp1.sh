# round 4, late: MSDA group -> workgroup maps (A/B), the pipelined fp32-feature mask contraction, the folded mask-features conv
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k "mask_einsum or (msda and not backward and not gradcheck)" 2>&1 | tail -4
timeout 150 python tools/bench_msda.py 2>&1 | tail -5
timeout 150 python tools/bench_einsum.py 2>&1 | head -3
timeout 400 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "full_size_split_policy and split3-e2e_full" 2>&1 | grep -v "^$" | tail -6
timeout 300 python bench.py --no-cpu-baseline --no-parity-leg 2>/dev/null | tail -1 | cut -c1-420
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-leg --timed-only > $GRAFT_REPO_ROOT/gpurun_out/x1_prof_timed.json 2> $GRAFT_REPO_ROOT/gpurun_out/x1_prof.err)
python tools/top_dispatches.py $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/x1_last5_forwards.txt 2>&1
head -40 gpurun_out/x1_last5_forwards.txt | cut -c1-150
