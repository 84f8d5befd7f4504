#!/bin/bash
# PMC passes (tools/pmc_kernel.sh) for the hand-written kernels at the bench geometry: the MFMA-bound global attention and the
# HBM-/gather-bound head kernels.  Output: gpurun_out/pmc_<tag>.txt, summarised by tools/pmc_summary.py into profiles/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
REL=2 DT=f16 bash tools/pmc_kernel.sh "python tools/bench_attn.py" vit_attn_sp_kernel:attention
bash tools/pmc_kernel.sh "python tools/bench_einsum.py" mask_einsum16_kernel:einsum16 dynamic_mask_mfma_kernel:dynmask16 dynamic_mask_kernel:dynmask
bash tools/pmc_kernel.sh "python tools/bench_msda.py" msda_d32_kernel:msda
DT=f16 bash tools/pmc_kernel.sh "python tools/bench_xattn.py" xattn_i2t_kernel:xattn_i2t xattn_t2i_kernel:xattn_t2i Li32:fa32
HIPIE_XATTN_GENERIC=1 DT=f16 bash tools/pmc_kernel.sh "python tools/bench_xattn.py" Li256:xattn256
