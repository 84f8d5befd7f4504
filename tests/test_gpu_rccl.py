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
