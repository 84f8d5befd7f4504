#!/bin/bash
# PMC counters of one or more kernels of ONE command, one rocprofv3 pass per counter set (SQ x2, GRBM, FETCH_SIZE,
# WRITE_SIZE), --kernel-trace only (never combined with sys/hip/hsa traces).
#   usage (on the GPU box, through gpurun):  bash tools/pmc_kernel.sh "<command>" <kernel substring>:<tag> [<substring>:<tag> ...]
#   e.g.  bash tools/pmc_kernel.sh "python tools/bench_einsum.py" mask_einsum_kernel:einsum dynamic_mask_kernel:dynmask
# Output: gpurun_out/pmc_<tag>.txt (per-launch averages, plus the mean kernel duration of the profiled launches).  FETCH_SIZE is
# reported as is (KiB): double it for 16-B/lane streaming reads before comparing with byte counts (MI355X_MICROARCH.md, HBM).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CMD="$1"; shift
mkdir -p $REPO/gpurun_out
for kt in "$@"; do : > $REPO/gpurun_out/pmc_${kt##*:}.txt; done
RUN=$(echo "$CMD" | md5sum | cut -c1-8)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  D=/tmp/pmc_${RUN}_$tag
  (cd $REPO && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D -o p -- $CMD > $D.log 2>&1)
  f=$(find $D -name "*counter_collection.csv" | head -1)
  for kt in "$@"; do
    python3 - "$f" "${kt%%:*}" >> $REPO/gpurun_out/pmc_${kt##*:}.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0.0, 0])
dur = [0.0, 0]
seen = set()
for r in rows:
    if sys.argv[2] not in r['Kernel_Name']: continue
    k = r['Counter_Name']; agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    did = r.get('Dispatch_Id')
    if did not in seen and r.get('End_Timestamp') and r.get('Start_Timestamp'):
        seen.add(did); dur[0] += float(r['End_Timestamp']) - float(r['Start_Timestamp']); dur[1] += 1
for k, (v, n) in sorted(agg.items()):
    print("PMC %-28s per-launch %.6g  (n=%d)" % (k, v / n, n))
if dur[1]:
    print("PMC %-28s per-launch %.6g us (n=%d, under this counter set)" % ("duration[" + sorted(agg)[0] + "]", dur[0] / dur[1] / 1e3, dur[1]))
PY
  done
done
for kt in "$@"; do echo "== ${kt##*:}"; cat $REPO/gpurun_out/pmc_${kt##*:}.txt; done
