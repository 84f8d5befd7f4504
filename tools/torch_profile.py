#!/usr/bin/env python3
"""torch.profiler view of one forward of the bench workload: device time grouped by (operator, input shapes), to find the
source of elementwise / copy / cast kernels.  Output: gpurun_out/torch_profile.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from hipie_amd.config import HipieConfig, Precision  # noqa: E402
from hipie_amd.hipie_img import HIPIE_IMG  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    cfg = HipieConfig.vit_huge()
    model = HIPIE_IMG(cfg, getattr(Precision, os.environ.get("POLICY", "split3"))(), device=dev)
    bench.randomize_degenerate_inits(model)
    model.finalize()
    batch = bench.synth_batch(cfg, 8, 1024, 80, 194, dev)
    for _ in range(2):
        model.forward_raw(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False, with_modules=False) as prof:
        model.forward_raw(batch)
        torch.cuda.synchronize()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/torch_profile.txt", "w") as f:
        f.write(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=90,
                                                                   max_name_column_width=40, max_shapes_column_width=90))
    print(open("gpurun_out/torch_profile.txt").read()[:200])
    if os.environ.get("STACKS") == "1":          # who issues the copies / casts: python stacks of the copy-like operators
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True,
                     experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof2:
            model.forward_raw(batch)
            torch.cuda.synchronize()
        import collections
        agg = collections.defaultdict(lambda: [0.0, 0])
        for e in prof2.events():
            if e.name in ("aten::copy_", "aten::add", "aten::masked_fill_", "aten::cat", "aten::mul", "aten::clamp", "aten::native_layer_norm",
                          "aten::add_", "aten::mul_", "aten::sigmoid", "aten::index", "aten::index_select", "aten::gather", "aten::where",
                          "aten::sub", "aten::div", "aten::relu", "aten::stack", "aten::softmax", "aten::_softmax", "aten::fill_", "aten::zero_") and e.device_time_total > 4:
                frames = [f for f in (e.stack or []) if "hipie_amd" in f]
                key = (e.name, str(e.input_shapes)[:60], " < ".join(f.split("/")[-1][:48] for f in frames[:2]) if frames else "?")
                agg[key][0] += e.device_time_total
                agg[key][1] += 1
        with open("gpurun_out/torch_profile_stacks.txt", "w") as f:
            for k, (t, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:120]:
                f.write("%8.1f us %4d x  %-16s %-62s %s\n" % (t, n, k[0], k[1], k[2]))


if __name__ == "__main__":
    main()
