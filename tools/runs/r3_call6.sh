cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 200 python tools/bench_attn_split.py > gpurun_out/c6_attn_bench.log 2>&1
timeout 900 bash tools/pmc_kernel.sh "python tools/bench_gemm_one.py qkv split" gemm_kernel:gemm_qkv_split > gpurun_out/c6_pmc_gemm.log 2>&1
timeout 900 bash tools/pmc_kernel.sh "python tools/bench_attn_split.py" vit_attn_split_kernelILi80ELi2ELi8:attn_split > gpurun_out/c6_pmc_attn.log 2>&1
python tools/pmc_summary.py gemm_qkv_split attn_split > gpurun_out/c6_pmc_summary.json 2>&1
