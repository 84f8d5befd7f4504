#!/bin/bash
# PMC counters of one kernel, one rocprofv3 pass per counter set (SQ x2, GRBM, FETCH_SIZE, WRITE_SIZE), --kernel-trace only.
#   usage (on the GPU box, through gpurun):  bash tools/pmc_kernel.sh "<command>" <kernel-name substring> <tag>
#   e.g.  bash tools/pmc_kernel.sh "python tools/bench_msda.py" msda_d32_kernel msda
# Output: gpurun_out/pmc_<tag>.txt (per-launch averages).  FETCH_SIZE is reported as is: double it for 16-B/lane streaming
# reads before comparing with byte counts (MI355X_MICROARCH.md, HBM section).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CMD="$1"; KERNEL="$2"; TAG="$3"
OUT=$REPO/gpurun_out/pmc_$TAG.txt
mkdir -p $REPO/gpurun_out
: > $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd $REPO && rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_${TAG}_$tag -o p -- $CMD > /tmp/pmc_${TAG}_$tag.log 2>&1)
  f=$(find /tmp/pmc_${TAG}_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$KERNEL" >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if sys.argv[2] not in r['Kernel_Name']: continue
    k = r['Counter_Name']; agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()):
    print("PMC %-28s per-launch %.6g  (n=%d)" % (k, v / n, n))
PY
done
cat $OUT
