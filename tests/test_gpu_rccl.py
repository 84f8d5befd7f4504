"""GPU: the data-parallel layer's collectives executed by RCCL (process group backend "nccl" IS RCCL on ROCm) on device tensors.

One GPU is what a gpurun box has, so the group has ONE rank (file:// rendezvous, hipie_amd.parallel.init_single_rank_group): the calls below
are the ones every rank of `bench.py --gpus N` makes per step -- all_gather_into_tensor of the fp32 prediction block and of the int16 map
block as bytes, the all-reduce behind live_ranks / max_over_ranks, the barrier -- on the backend the N > 1 job uses (reference equivalent:
detectron2/utils/comm.py:87-153 pickle gather over gloo, engine/launch.py:98-117).  The world-size-2 semantics are covered on gloo by
tests/test_dp_gloo.py; the 8-GPU curve is the driver's to measure.  Runs in a child process with a hard timeout so that a wedged
communicator cannot take the test session with it.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from hipie_amd import parallel
backend = parallel.init_single_rank_group("nccl", 0)
assert backend == "nccl", backend
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
block = torch.randn(8, 100, parallel.PRED_FIELDS, generator=g).to(dev)
maps = torch.randint(-1, 150, (8, 2, 256, 256), generator=g).to(torch.int16).to(dev)
out = parallel.all_gather_predictions(block)
mout = parallel.all_gather_predictions(maps)
torch.cuda.synchronize()
assert out.data_ptr() != block.data_ptr() and torch.equal(out, block)            # a real gather into a new buffer, not the identity short-cut
assert mout.dtype == torch.int16 and mout.data_ptr() != maps.data_ptr() and torch.equal(mout, maps)
ranks = parallel.live_ranks(dev)
t = parallel.max_over_ranks(1.25, dev)
parallel.barrier()
dp = parallel.dp_evidence(out, 8, 0, 1, dev)
# the same through a side stream while the default stream is busy (the step issues the gather right after its last kernel)
a = torch.randn(4096, 4096, device=dev)
for _ in range(4):
    a = a @ a * 1e-3
out2 = parallel.all_gather_predictions(block * 2)
torch.cuda.synchronize()
assert torch.equal(out2, block * 2)
# training side: the bucketed gradient all-reduce on the same backend (one rank: the mean is the gradient itself, but the collectives run)
from hipie_amd.training import GradientBuckets
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 256)).to(dev)
x = torch.randn(64, 256, device=dev)
torch.set_grad_enabled(True)
m(x).square().sum().backward()
want = [p.grad.clone() for p in m.parameters()]
m.zero_grad(set_to_none=True)
gb = GradientBuckets(m.parameters(), bucket_mb=0.3)
m(x).square().sum().backward()
n_allreduce = gb.finish()
torch.cuda.synchronize()
assert n_allreduce == len(gb.buckets) >= 2
for p, w in zip(m.parameters(), want):
    assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-6)
parallel.shutdown()
print("RCCL_OK " + json.dumps({"ranks": ranks, "t": t, "dp": dp, "world": 1}))
"""


def test_collectives_run_through_rccl_on_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=240, env=env)
    tag = [l for l in r.stdout.splitlines() if l.startswith("RCCL_OK ")]
    assert r.returncode == 0 and tag, (r.stdout[-2000:], r.stderr[-2000:])
    import json
    res = json.loads(tag[-1][len("RCCL_OK "):])
    assert res["ranks"] == 1 and res["t"] == 1.25
    assert res["dp"]["backend"] == "nccl" and res["dp"]["rccl_ranks"] == 1 and res["dp"]["gathered_block_shape"] == [8, 100, 7]


SHARD_CHILD = r"""
import json, os, sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
sys.path.insert(0, os.path.join(sys.path[0], "golden"))
import _synth
from util import Golden
from hipie_amd import parallel
from hipie_amd.config import HipieConfig, Precision
from hipie_amd.hipie_img import HIPIE_IMG
from hipie_amd.postprocess import inference_compact
torch.set_grad_enabled(False)
rank, world, _ = parallel.init_from_env(backend="gloo")          # two processes, ONE GPU: the group runs on gloo, the compute on cuda:0
dev = torch.device("cuda", 0)
g = Golden("e2e_tiny")
model = HIPIE_IMG(HipieConfig.from_dict(g.meta["cfg"]), Precision.split3(), device=dev)
model.load_state_dict(_synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, dist=g.meta.get("dist")), strict=True)
model.finalize()
total = 4
imgs = _synth.synth_images([(192, 256)] * total, seed=99)
ids, mask, pmap = _synth.synth_token_ids(total, 9, 64, seed=74)
def batch(idx):
    return [{"image": imgs[i], "task": "detection", "input_ids": ids[0], "attention_mask": mask[0], "positive_map_label_to_token": pmap,
             "height": 192, "width": 256} for i in idx]
everything = batch(range(total))
full_out = model.forward_raw(everything)                         # the whole global batch in this process: the yardstick, and the top-k rows to pin
fg, md = model.last_topk()
full = inference_compact(model, full_out, everything, topk=20)
shard = parallel.shard_range(total, rank, world)                 # InferenceSampler ranges: rank 0 -> images 0, 1; rank 1 -> images 2, 3
model.pin_topk(fg[shard.start:shard.stop].cpu(), md[shard.start:shard.stop].cpu())
mine = batch(shard)
local = inference_compact(model, model.forward_raw(mine), mine, topk=20)
gathered = parallel.all_gather_predictions(local)
t = parallel.max_over_ranks(1.0 + rank, dev)
dp = parallel.dp_evidence(gathered, len(shard), rank, world, dev)
parallel.barrier()
same = (gathered[..., 5:] == full[..., 5:]).all(-1)             # rows whose (class, query index) agree: a near-tie of two scores may swap a pair
boxes_err = float(((gathered[..., :5] - full[..., :5]).abs() * same[..., None]).max() / full[..., :5].abs().max())
same_ids = float(same.float().mean())
print("SHARD_OK " + json.dumps({"rank": rank, "shape": list(gathered.shape), "boxes_err": boxes_err, "same_ids": same_ids, "t": t, "dp": dp,
                                "device": str(gathered.device), "n_inst": int((full[..., 4] > 0).sum())}))
parallel.shutdown()
"""


def test_two_processes_shard_the_batch_on_device_tensors():
    """N > 1 on DEVICE tensors with the hardware a gpurun box has: two processes share cuda:0 (RCCL refuses two ranks on one GPU, so the group
    runs on gloo and the blocks cross through the host -- parallel.all_gather_predictions), each runs the product forward + the device-side
    compact predictions on ITS contiguous shard of a 4-image global batch (parallel.shard_range == InferenceSampler._get_local_indices,
    detectron2/data/samplers/distributed_sampler.py:245-278) and all-gathers; every rank must end up with the block one process computes
    for the whole batch (class / query ids equal up to swaps of near-tied scores, boxes and scores within the batch-independence bound
    of test_gpu_e2e)."""
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, "-c", SHARD_CHILD % (ROOT, ROOT)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    res = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=400)
            tag = [l for l in out.splitlines() if l.startswith("SHARD_OK ")]
            assert p.returncode == 0 and tag, (out[-2000:], err[-3000:])
            res.append(json.loads(tag[-1][len("SHARD_OK "):]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res.sort(key=lambda r: r["rank"])
    for r in res:
        assert r["shape"] == [4, 20, 7] and r["device"].startswith("cuda") and r["n_inst"] > 0
        assert r["same_ids"] >= 0.95 and r["boxes_err"] < 1e-3, r
        assert r["t"] == 2.0 and r["dp"]["rccl_ranks"] == 2 and r["dp"]["backend"] == "gloo" and r["dp"]["global_images"] == 4
    assert res[0]["dp"]["shard_of_this_rank"] == [0, 2] and res[1]["dp"]["shard_of_this_rank"] == [2, 4]
