# round 4, the last GPU seconds: one SQ counter pass over the K = 256 projection (N = 256, fp32 rows) on the tile kernel and on the thin-K kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cat > /tmp/k256_few.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from hipie_amd import ops
x = torch.randn(174080, 256, device="cuda"); w = ops.hl8_pack(torch.randn(256, 256, device="cuda") / 16); b = torch.randn(256, device="cuda")
for mode in ("0", "1"):
    os.environ["HIPIE_GEMM_K256"] = mode
    for _ in range(6):
        ops.gemm(x, w, b, out_fmt=ops.F32, split=True)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pk -o p -- python /tmp/k256_few.py > /tmp/pk.log 2>&1
python3 - $(find /tmp/pk -name "*counter_collection.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_k256_sq.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm_k" not in k: continue
    name = "thin (gemm_k256_kernel)" if "k256" in k else "tile (" + k.split("(")[0][-28:] + ")"
    a = agg[name][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for name, c in agg.items():
    wc = c["SQ_WAVE_CYCLES"][0] / c["SQ_WAVE_CYCLES"][1]
    print(name, "launches", c["SQ_WAVE_CYCLES"][1])
    for k, (v, n) in sorted(c.items()):
        print("   %-26s %.4g  (%.3f of wave cycles)" % (k, v / n, v / n / wc))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_k256_sq.txt; tail -3 /tmp/pk.log
