# round 4, late: PMC passes of the HBM- / gather-bound head kernels after this round's changes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIPIE_MIOPEN_FIND=0
timeout 800 bash tools/pmc_kernel.sh "python tools/bench_hbm_kernels.py" mask_einsum_dma_kernel:einsum_dma msda_d32_kernel:msda_tile dynamic_mask_mfma_kernel:dynmask_split > gpurun_out/p_pmc_hbm.log 2>&1
python tools/pmc_summary.py einsum_dma msda_tile dynmask_split > gpurun_out/r04_pmc_hbm_kernels.json 2> gpurun_out/p_pmc_hbm_summary.err
cat gpurun_out/pmc_einsum_dma.txt gpurun_out/pmc_msda_tile.txt gpurun_out/pmc_dynmask_split.txt > gpurun_out/r04_pmc_hbm_raw.txt
cat gpurun_out/r04_pmc_hbm_kernels.json | head -80; tail -3 gpurun_out/p_pmc_hbm_summary.err
