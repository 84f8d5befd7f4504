#!/bin/bash
# PMC passes (tools/pmc_kernel.sh) for the HBM-/gather-bound hand-written kernels at the bench geometry.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/pmc_kernel.sh "python tools/bench_einsum.py" mask_einsum_kernel:einsum dynamic_mask_kernel:dynmask
bash tools/pmc_kernel.sh "python tools/bench_msda.py" msda_d32_kernel:msda
bash tools/pmc_kernel.sh "python tools/bench_xattn.py" Li256:xattn256 Li32:fa32
