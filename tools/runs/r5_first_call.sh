# round 5, suggested first GPU call (~6 min): where the non-ViT third of the step goes on the round-4 tree, and the A/B switches that exist
#   - stage / kernel-class table of one forward (shapes included), per-kernel stats of the timed steps
#   - SQ counter passes of the three latency-bound hand-written kernels (K = 256 GEMM tile + thin form, fused FFN, mask contraction)
#   - the env switches of round 4, each as a short bench line: HIPIE_GEMM_K256=0|1 (default: N >= 384), HIPIE_MSDA_MAP=0 (heads-fastest),
#     HIPIE_ME_XMAP=0 (batch-major mask-contraction ids), HIPIE_GEMM_SMALL=0
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 300 python tools/stage_times.py split3 shapes > gpurun_out/r05_stage_times.txt 2>&1; head -45 gpurun_out/r05_stage_times.txt
timeout 200 python tools/bench_gemm_k256.py 2>&1 | tail -8
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_ffn_fused.py fused" ffn_fused_kernel:r05_ffn_fused > gpurun_out/r05_pmc_ffn.log 2>&1
timeout 600 bash tools/pmc_kernel.sh "python tools/bench_hbm_kernels.py" mask_einsum_dma_kernel:r05_einsum msda_d32_kernel:r05_msda > gpurun_out/r05_pmc_hbm.log 2>&1
python tools/pmc_summary.py r05_ffn_fused r05_einsum r05_msda > gpurun_out/r05_pmc_kernels.json 2>/dev/null; head -60 gpurun_out/r05_pmc_kernels.json
for sw in "" "HIPIE_GEMM_K256=0" "HIPIE_GEMM_K256=1" "HIPIE_MSDA_MAP=0" "HIPIE_ME_XMAP=0"; do
  echo "== $sw"; env $sw timeout 300 python bench.py --no-cpu-baseline --no-parity-leg 2>/dev/null | tail -1 | cut -c1-260
done
