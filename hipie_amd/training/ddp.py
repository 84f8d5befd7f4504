"""Data-parallel gradient averaging for the training side: what detectron2's create_ddp_model
(detectron2/engine/defaults.py:60-79, torch DistributedDataParallel) does for the reference -- one process per GPU, gradients
averaged over the ranks while the backward pass is still running -- written for this node: RCCL over xGMI is a ring per link
(7 links x ~153 GB/s), so the buckets are few and large (64 MB by default, the ViT-H blocks give ~40 of them) instead of DDP's 25 MB
first-bucket-1 MB schedule for NVSwitch, and the flat bucket IS the gradient storage (param.grad are views into it: no copy in, no
copy out).  `fp16_compression` = the reference's optional comm hook (fp16 on the wire, fp32 accumulate on arrival)."""
import torch
import torch.distributed as dist


class GradientBuckets:
    """usage:  gb = GradientBuckets(model.parameters());  loss.backward();  gb.finish();  optimizer.step();  gb.zero_grad()

    Parameters are bucketed in REVERSE registration order (the order autograd produces their gradients in); a bucket's all-reduce is
    launched asynchronously from the hook of the last parameter of the bucket to receive its gradient.  finish() waits for the
    outstanding collectives and divides by the world size (launching any bucket a skipped parameter left incomplete).  Without an
    initialised process group everything degrades to a no-op average over one rank."""

    def __init__(self, params, bucket_mb=64.0, fp16_compression=False, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.fp16 = bool(fp16_compression)
        self.buckets = []                   # (flat, [params], pending set of ids)
        cap = int(bucket_mb * (1 << 20))
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._work = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(bi)) for bi, (_, ps, _) in enumerate(self.buckets) for p in ps]

    def _close(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)        # the gradient accumulates straight into the bucket
            off += p.numel()
        self.buckets.append((flat, ps, set()))

    def _make_hook(self, bi):
        def hook(p):
            flat, ps, seen = self.buckets[bi]
            seen.add(id(p))
            if len(seen) == len(ps):
                self._launch(bi)
        return hook

    def _launch(self, bi):
        flat, ps, seen = self.buckets[bi]
        seen.clear()
        seen.add("launched")
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1 and dist.get_backend(self.group) != "nccl":
            return
        for p in ps:                                             # a hook-less path may have replaced .grad: keep the views authoritative
            if p.grad is not None and p.grad.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr():
                raise RuntimeError("GradientBuckets: a parameter's .grad was re-allocated outside its bucket")
        if self.fp16 and flat.dtype == torch.float32:
            wire = flat.half()
            self._work.append((dist.all_reduce(wire, group=self.group, async_op=True), flat, wire))
        else:
            self._work.append((dist.all_reduce(flat, group=self.group, async_op=True), flat, None))

    def finish(self):
        """wait for every bucket, average.  Returns the number of all-reduces that ran."""
        for bi, (flat, ps, seen) in enumerate(self.buckets):
            if "launched" not in seen:                           # some parameter of the bucket got no gradient this step
                self._launch(bi)
        n = len(self._work)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        for work, flat, wire in self._work:
            work.wait()
            if wire is not None:
                flat.copy_(wire)
            if world > 1:
                flat.div_(world)
        self._work = []
        for _, _, seen in self.buckets:
            seen.clear()
        return n

    def zero_grad(self):
        for flat, _, _ in self.buckets:
            flat.zero_()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
