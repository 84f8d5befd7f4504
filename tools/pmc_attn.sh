#!/bin/bash
# PMC counters for the ViT global-attention kernel with the in-kernel rel-pos bias (run on the GPU box through gpurun).
# Every counter set is its own pass, with --kernel-trace only (no sys/hip/hsa traces).  Output: gpurun_out/pmc_attn.txt
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_attn.txt
mkdir -p $REPO/gpurun_out
: > $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  FUSED_ONLY=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$tag -o p -- python $REPO/tools/bench_attn.py > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if 'flash_attn' not in r['Kernel_Name']: continue
    k = r['Counter_Name']; agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()):
    print("PMC %-28s per-launch %.6g  (n=%d)" % (k, v / n, n))
PY
done
cat $OUT
