#!/usr/bin/env python3
"""one-off: let PyTorch TunableOp pick the hipBLASLt / rocBLAS solution for every GEMM shape of the bench forward
(ViT-H, 1024^2, batch 8, the timed fp16 policy).  The table is written at interpreter exit to gpurun_out/tunableop_full.csv; copy it
to hipie_amd/tuning/tunableop_gfx950_vith_bs8.csv to have bench.py use it (read-only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from hipie_amd.config import HipieConfig, Precision  # noqa: E402
from hipie_amd.hipie_img import HIPIE_IMG  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    os.makedirs("gpurun_out", exist_ok=True)
    committed = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hipie_amd", "tuning", "tunableop_gfx950_vith_bs8.csv")
    if os.path.exists(committed) and os.environ.get("FRESH") != "1":      # seed: only shapes that are not in the committed table get tuned
        import shutil
        shutil.copyfile(committed, "gpurun_out/tunableop_full.csv")
    torch.cuda.tunable.set_filename("gpurun_out/tunableop_full.csv")
    torch.cuda.tunable.set_max_tuning_duration(15)
    torch.cuda.tunable.set_max_tuning_iterations(10)
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    dev = torch.device("cuda", 0)
    cfg = HipieConfig.vit_huge()
    model = HIPIE_IMG(cfg, Precision.fast(), device=dev)
    bench.randomize_degenerate_inits(model)
    model.finalize()
    batch = bench.synth_batch(cfg, 8, 1024, 80, 194, dev)
    for _ in range(2):
        model.forward_raw(batch)
    torch.cuda.synchronize()
    print("tuned entries:", len(torch.cuda.tunable.get_results()))


if __name__ == "__main__":
    main()
