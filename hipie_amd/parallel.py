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


def init_single_rank_group(backend=None, device_index=0):
    """a process group of ONE rank (file:// rendezvous, no port): the N = 1 step then runs the same collectives as the N > 1 one --
    all_gather_into_tensor / all_reduce go through RCCL when the backend is "nccl" -- instead of short-cutting them.  Returns the
    backend name, or None when the group could not be created (the collectives then degrade to identities, as without a group)."""
    import tempfile
    if dist.is_initialized():
        return dist.get_backend()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(device_index)
    path = os.path.join(tempfile.gettempdir(), "hipie_pg_%d_%d" % (os.getpid(), int.from_bytes(os.urandom(4), "little")))
    try:
        dist.init_process_group(backend=backend, init_method="file://" + path, rank=0, world_size=1)
    except Exception as e:          # pragma: no cover - depends on the box
        import sys
        print("hipie_amd.parallel: single-rank %s group not available (%r); running without a process group" % (backend, e), file=sys.stderr)
        return None
    return dist.get_backend()


def init_from_env(backend=None, single_rank_group=False):
    """torchrun-style env (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world, local_rank).
    single_rank_group: with WORLD_SIZE 1 also open a one-rank process group (init_single_rank_group), so the collectives really run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and single_rank_group and not dist.is_initialized():
        init_single_rank_group(backend, local)
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
        block[i, :k, 6] = inst.query_index[:k].float() if inst.has("query_index") else -1.0
    return block


def compact_maps(results, hw, stride=4, device=None):
    """per-image semantic and panoptic label maps at 1/stride resolution: (n_img, 2, hw[0], hw[1]) int16,
    channel 0 = arg-max class of `sem_seg` (C,H,W), channel 1 = panoptic segment id of `panoptic_seg[0]` (H,W); -1 where the
    image is smaller than the block or the map is absent (task "grounding", or with_sem_pan=False).  Nearest subsampling."""
    n = len(results)
    dev = device or results[0]["instances"].scores.device
    block = torch.full((n, 2, hw[0], hw[1]), -1, dtype=torch.int16, device=dev)
    for i, r in enumerate(results):
        sem, pan = r.get("sem_seg"), (r.get("panoptic_seg") or (None, None))[0]
        if sem is not None:
            m = sem[:, ::stride, ::stride].argmax(0)[:hw[0], :hw[1]]
            block[i, 0, :m.shape[0], :m.shape[1]] = m.to(torch.int16)
        if pan is not None:
            m = pan[::stride, ::stride][:hw[0], :hw[1]]
            block[i, 1, :m.shape[0], :m.shape[1]] = m.to(torch.int16)
    return block


def all_gather_predictions(block):
    """(n_local, ...) per rank (same shape on every rank: pad upstream) -> (world*n_local, ...) on every rank.  One
    all_gather_into_tensor per block: the (n, topk, 7) fp32 instance block and, when computed, the (n, 2, h, w) int16 map block."""
    if not dist.is_initialized():
        return block
    world = dist.get_world_size()
    block = block.contiguous()
    raw = block if block.dtype == torch.float32 else block.view(torch.uint8)      # bytes: every backend gathers uint8
    if raw.is_cuda and dist.get_backend() == "gloo":
        # gloo is the CPU backend of the tests (and of N processes sharing fewer GPUs, tests/test_gpu_rccl.py): device blocks cross
        # through the host there; the N-GPU job's backend is "nccl" (RCCL), which gathers device memory directly
        host = torch.empty((world * raw.shape[0],) + tuple(raw.shape[1:]), dtype=raw.dtype)
        dist.all_gather_into_tensor(host, raw.cpu())
        out = host.to(raw.device)
    else:
        out = torch.empty((world * raw.shape[0],) + tuple(raw.shape[1:]), dtype=raw.dtype, device=raw.device)
        dist.all_gather_into_tensor(out, raw)
    return out if block.dtype == torch.float32 else out.view(block.dtype)


def _collective_device(device):
    """where the small bookkeeping collectives (rank count, max-over-ranks) put their operand: the device for "nccl", the host for "gloo"."""
    return torch.device("cpu") if dist.is_initialized() and dist.get_backend() == "gloo" else device


def live_ranks(device):
    """number of ranks that actually take part in the process group's collectives: an all-reduce (sum) of ones over the backend
    in use (RCCL for "nccl").  1 without a process group.  bench.py reports it next to the gathered block's shape."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.float32, device=_collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def backend_name():
    return dist.get_backend() if dist.is_initialized() else None


def dp_evidence(gathered, per_rank, rank, world, device):
    """what bench.py prints about the data-parallel step: `rccl_ranks` = the ranks the process group's collectives really see (an
    all-reduce of ones; RCCL when the backend is "nccl"), the backend, the global batch and this rank's contiguous shard of it
    (shard_range), and the shape of the block every rank holds after the all-gather."""
    shard = shard_range(per_rank * world, rank, world)
    return {"rccl_ranks": live_ranks(device), "backend": backend_name(), "global_images": per_rank * world,
            "shard_of_this_rank": [shard.start, shard.stop], "gathered_block_shape": None if gathered is None else list(gathered.shape)}


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_collective_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
