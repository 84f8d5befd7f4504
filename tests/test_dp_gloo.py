"""CPU, world_size 2 over gloo: the data-parallel layer (sharding, fixed-shape all-gather, max-over-ranks timing)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from hipie_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    idx = list(parallel.shard_range(total, r, w))
    n_local = (total + w - 1) // w
    results = []
    for i in range(n_local):
        gi = idx[i] if i < len(idx) else -1
        k = 3
        from hipie_amd.structures import Boxes, Instances
        sem = torch.zeros(3, 8, 8)
        sem[max(gi, 0) % 3] = 1.0                                  # arg-max class = image index mod 3
        results.append({"instances": Instances((8, 8), pred_boxes=Boxes(torch.full((k, 4), float(gi))),
                                               scores=torch.full((k,), float(gi) + 0.5),
                                               pred_classes=torch.full((k,), gi, dtype=torch.long),
                                               query_index=torch.tensor([7, 3, 11], dtype=torch.int32) + max(gi, 0)),
                        "sem_seg": sem if gi >= 0 else None,
                        "panoptic_seg": (torch.full((8, 8), gi + 1, dtype=torch.int32), []) if gi >= 0 else (None, None)})
    block = parallel.compact_predictions(results, topk=5)
    out = parallel.all_gather_predictions(block)
    maps = parallel.all_gather_predictions(parallel.compact_maps(results, (2, 2), stride=4))
    t = parallel.max_over_ranks(1.0 + r, torch.device("cpu"))
    dp = parallel.dp_evidence(out, n_local, r, w, torch.device("cpu"))      # what bench.py --gpus N prints as "dp"
    parallel.barrier()
    q.put((r, idx, out.tolist(), t, maps.tolist(), dp))     # plain lists: no shared-memory handles across process exit


def test_two_rank_shard_and_gather():
    world, total = 2, 7
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][1] == [0, 1, 2, 3] and got[1][1] == [4, 5, 6]           # contiguous ranges, first rank takes the extra
    assert got[0][2] == got[1][2]                                          # every rank holds the full gather
    full = torch.tensor(got[0][2])
    assert full.shape == (8, 5, 7)
    seen = sorted(int(v) for v in full[:, 0, 5].tolist())
    assert seen == [-1, 0, 1, 2, 3, 4, 5, 6]                               # 7 images + one pad slot
    assert got[0][3] == 2.0 and got[1][3] == 2.0                           # max over ranks
    img3 = [i for i in range(8) if full[i, 0, 5] == 3][0]
    assert full[img3, :3, 6].tolist() == [10.0, 6.0, 14.0] and full[img3, 3, 6] == 0.0      # real query indices, zero padding
    assert got[0][4] == got[1][4]
    maps = torch.tensor(got[0][4])
    assert maps.shape == (8, 2, 2, 2)
    assert (maps[img3, 0] == 0).all() and (maps[img3, 1] == 4).all()        # class 3 % 3, panoptic id 3 + 1
    pad = [i for i in range(8) if full[i, 0, 5] == -1][0]
    assert (maps[pad] == -1).all()                                          # the padded slot carries no maps
    # the self-evidence fields of bench.py's N > 1 line: ranks seen by an all-reduce of ones, and the gathered block's shape
    for r in (0, 1):
        dp = got[r][5]
        assert dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and dp["global_images"] == 8
        assert dp["gathered_block_shape"] == [8, 5, 7]
    assert got[0][5]["shard_of_this_rank"] == [0, 4] and got[1][5]["shard_of_this_rank"] == [4, 8]


def test_shard_range_covers_everything():
    from hipie_amd.parallel import shard_range
    for total in (0, 1, 8, 64, 65):
        for world in (1, 2, 8):
            allidx = [i for r in range(world) for i in shard_range(total, r, world)]
            assert allidx == list(range(total))


def _single(q):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    from hipie_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo", single_rank_group=True)          # what bench.py does at N = 1 (there: "nccl")
    block = torch.arange(2 * 3 * parallel.PRED_FIELDS, dtype=torch.float32).view(2, 3, -1)
    out = parallel.all_gather_predictions(block)
    maps = torch.arange(2 * 2 * 4 * 4, dtype=torch.int16).view(2, 2, 4, 4)
    mout = parallel.all_gather_predictions(maps)
    dp = parallel.dp_evidence(out, 2, r, w, torch.device("cpu"))
    t = parallel.max_over_ranks(0.5, torch.device("cpu"))
    parallel.barrier()
    parallel.shutdown()
    q.put((r, w, out.data_ptr() != block.data_ptr(), torch.equal(out, block), torch.equal(mout, maps), dp, t))


def test_single_rank_group_runs_the_collectives():
    """N = 1 with a one-rank process group (file:// rendezvous): the gathers and reductions go through the backend instead of being
    short-cut -- the code path bench.py --gpus 1 times (backend "nccl" there: tests/test_gpu_rccl.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single, args=(q,))
    p.start()
    r, w, fresh, same, msame, dp, t = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    assert (r, w) == (0, 1) and fresh and same and msame and t == 0.5
    assert dp["backend"] == "gloo" and dp["rccl_ranks"] == 1 and dp["gathered_block_shape"] == [2, 3, 7]


# --------------------------------------------------------------------------- training side: bucketed gradient all-reduce (row f-4)
def _toy_model():
    torch.manual_seed(5)
    m = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    m.unused = torch.nn.Parameter(torch.ones(4))            # receives no gradient: its bucket is launched by finish()
    return m


def _toy_grads(rank):
    m = _toy_model()
    x = torch.randn(9, 6, generator=torch.Generator().manual_seed(100 + rank))
    with torch.enable_grad():                               # the suite runs with grad mode off (conftest)
        m(x).square().sum().backward()
    return [None if p.grad is None else p.grad.clone() for p in m.parameters()]


def _ddp_worker(rank, world, port, fp16, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from hipie_amd import parallel
    from hipie_amd.training import GradientBuckets
    parallel.init_from_env(backend="gloo")
    m = _toy_model()
    gb = GradientBuckets(m.parameters(), bucket_mb=600 / (1 << 20), fp16_compression=fp16)       # 600-byte buckets: several of them
    outs = []
    for step in range(2):                                   # second step: zero_grad + the same hooks again
        x = torch.randn(9, 6, generator=torch.Generator().manual_seed(100 + rank))
        with torch.enable_grad():
            m(x).square().sum().backward()
        n = gb.finish()
        outs.append([p.grad.clone().tolist() for p in m.parameters()])
        gb.zero_grad()
    parallel.barrier()
    q.put((rank, len(gb.buckets), n, outs))


def _run_ddp(fp16):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ddp_worker, args=(r, world, port, fp16, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return got


def test_gradient_buckets_average_over_two_ranks():
    """GradientBuckets (detectron2/engine/defaults.py:60-79 create_ddp_model's job): after backward + finish() every rank holds the MEAN of
    the two ranks' gradients, for every bucket, on both steps; the parameter without a gradient stays zero and does not hang its bucket."""
    got = _run_ddp(False)
    g0, g1 = _toy_grads(0), _toy_grads(1)
    assert got[0][1] > 2 and got[0][2] == got[0][1]                        # several buckets, one all-reduce each
    for step in range(2):
        for a, b, x0, x1 in zip(got[0][3][step], got[1][3][step], g0, g1):
            a, b = torch.tensor(a), torch.tensor(b)
            assert torch.equal(a, b)
            want = torch.zeros_like(a) if x0 is None else (x0 + x1) / 2
            assert torch.allclose(a, want, rtol=1e-6, atol=1e-7)


def test_gradient_buckets_fp16_wire():
    """the reference's optional fp16_compress_hook: fp16 on the wire, averages within fp16 rounding of the fp32 mean."""
    got = _run_ddp(True)
    g0, g1 = _toy_grads(0), _toy_grads(1)
    for a, x0, x1 in zip(got[0][3][0], g0, g1):
        a = torch.tensor(a)
        want = torch.zeros_like(a) if x0 is None else (x0 + x1) / 2
        assert torch.allclose(a, want, rtol=2e-3, atol=1e-3 * float(want.abs().max() + 1e-6))


def _ddp_uneven_worker(rank, world, port, q):
    """rank 1 never uses the middle Linear (a data-dependent branch); large gradients on the fp16 wire; a second backward is refused"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from hipie_amd import parallel
    from hipie_amd.training import GradientBuckets
    parallel.init_from_env(backend="gloo")
    m = _toy_model()
    gb = GradientBuckets(m.parameters(), bucket_mb=600 / (1 << 20))
    x = torch.randn(9, 6, generator=torch.Generator().manual_seed(100 + rank))
    with torch.enable_grad():
        h = m[1](m[0](x))
        if rank == 0:
            h = m[3](m[2](h))
        m[4](h).square().sum().backward()
    n = gb.finish()
    grads = [p.grad.clone().tolist() for p in m.parameters()]
    gb.zero_grad()
    refused = False
    with torch.enable_grad():
        m(x).square().sum().backward()
        try:
            m(x).square().sum().backward()                   # a second backward before finish(): the buckets are already on the wire
        except RuntimeError:
            refused = True
    gb.finish()
    gb.remove()
    big = torch.nn.Linear(4, 4)
    gb2 = GradientBuckets(big.parameters(), fp16_compression=True)
    with torch.enable_grad():
        (big(torch.ones(2, 4)).sum() * 3e4).backward()       # gradients of 6e4: the SUM over two ranks overflows fp16, the mean does not
    gb2.finish()
    finite = bool(all(torch.isfinite(p.grad).all() for p in big.parameters()))
    top = float(max(p.grad.abs().max() for p in big.parameters()))
    parallel.barrier()
    q.put((rank, n, len(gb.buckets), grads, refused, finite, top))


def test_gradient_buckets_uneven_participation_and_wire_range():
    """(1) ranks whose unused-parameter sets differ issue the SAME collectives in the SAME order (no hang, mean gradients, zero
    contribution from the rank that skipped the layer); (2) a second backward before finish() is refused instead of reducing a bucket
    twice; (3) the fp16 wire divides before the reduce: 6e4-sized gradients stay finite."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ddp_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][1] == got[0][2] == got[1][1]               # every bucket reduced once on both ranks
    for a, b in zip(got[0][3], got[1][3]):
        assert torch.equal(torch.tensor(a), torch.tensor(b))
    m = _toy_model()
    x0 = torch.randn(9, 6, generator=torch.Generator().manual_seed(100))
    with torch.enable_grad():
        m(x0).square().sum().backward()
    names = [n for n, _ in m.named_parameters()]
    mid = torch.tensor(got[0][3][names.index("2.weight")])    # weight of the middle Linear: only rank 0 contributed
    assert torch.allclose(mid, m[2].weight.grad / 2, rtol=1e-6, atol=1e-7)
    assert got[0][4] and got[1][4] and got[0][5] and got[1][5] and 5e4 < got[0][6] < 7e4


def _train_dp_worker(rank, world, port, q):
    """two ranks, one image each of the training fixture: train_iteration with GradientBuckets; both ranks must end with the same weights"""
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_training as TT
    from hipie_amd import parallel
    from hipie_amd.training import GradientBuckets, build_optimizer, train_iteration
    parallel.init_from_env(backend="gloo")
    torch.set_num_threads(2)
    z, meta, model, step, batch, targets = TT._train_step_case("cpu")
    for p in model.text_encoder.parameters():
        p.requires_grad_(False)
    opt = build_optimizer(model)
    gb = GradientBuckets([p for g in opt.param_groups for p in g["params"]], bucket_mb=8.0)
    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    losses, norm = train_iteration(step, opt, batch[rank:rank + 1], targets[rank:rank + 1], buckets=gb)
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters() if n in before)
    digest = float(sum(p.detach().double().sum() for n, p in model.named_parameters() if n in before))
    parallel.barrier()
    q.put((rank, float(sum(losses.values())), float(norm), moved, len(before), digest))


def test_training_iteration_data_parallel_two_ranks():
    """row f-4 + (e): the TRAINING iteration under data parallelism on two gloo ranks (what create_ddp_model + SimpleTrainer.run_step do for
    the reference, detectron2/engine/defaults.py:60-79): each rank runs the step on its own image, the gradients are averaged bucket by
    bucket during backward, clipped, AdamW steps -- the ranks see different losses, the SAME gradient norm, and end with identical weights."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_train_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=300) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (r0, l0, n0, m0, k0, d0), (r1, l1, n1, m1, k1, d1) = got
    assert abs(l0 - l1) > 1e-3 * abs(l0)                      # different images, different losses
    assert abs(n0 - n1) <= 1e-5 * n0 and n0 > 0               # one averaged gradient on both ranks
    assert m0 == m1 and m0 > 0.95 * k0                        # (nearly) every trainable tensor moved
    assert d0 == d1                                           # bit-identical weights after the step
