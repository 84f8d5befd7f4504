"""Data-parallel inference over the GPUs of one node: one process per GPU, images sharded in contiguous
index ranges exactly like detectron2's InferenceSampler (D2/data/samplers/distributed_sampler.py:265-272), no exchange
during the forward, ONE fixed-shape tensor all-gather of compact per-image predictions at the end (RCCL over xGMI when
the process group backend is "nccl"; "gloo" on CPU for the tests).  Replaces the reference's pickle gather over a gloo
group (D2/utils/comm.py:87-153).
"""
import os

import torch
import torch.distributed as dist

PRED_FIELDS = 7          # x0, y0, x1, y1, score, class, query index


def init_from_env(backend=None):
    """torchrun-style env (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total, rank, world):
    """InferenceSampler._get_local_indices: the first (total % world) ranks get one extra item."""
    shard = total // world
    left = total % world
    sizes = [shard + int(r < left) for r in range(world)]
    begin = sum(sizes[:rank])
    return range(begin, min(begin + sizes[rank], total))


def compact_predictions(results, topk=100, device=None):
    """list of per-image result dicts (postprocess.inference) -> (n_img, topk, PRED_FIELDS) float32 block, zero padded."""
    n = len(results)
    dev = device or results[0]["instances"].scores.device
    block = torch.zeros(n, topk, PRED_FIELDS, dtype=torch.float32, device=dev)
    for i, r in enumerate(results):
        inst = r["instances"]
        k = min(topk, inst.scores.shape[0])
        block[i, :k, :4] = inst.pred_boxes.tensor[:k]
        block[i, :k, 4] = inst.scores[:k]
        block[i, :k, 5] = inst.pred_classes[:k].float()
        block[i, :k, 6] = torch.arange(k, device=dev, dtype=torch.float32)
    return block


def all_gather_predictions(block):
    """(n_local, topk, F) per rank (same n_local on every rank: pad upstream) -> (world*n_local, topk, F) on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return block
    world = dist.get_world_size()
    out = torch.empty((world * block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, block.contiguous())
    return out


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
