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

    Parameters are bucketed in REVERSE registration order (the order autograd produces their gradients in).  A bucket becomes READY when
    the last of its parameters has received its gradient; collectives are ISSUED strictly in bucket order -- bucket i only once buckets
    0 .. i-1 are issued -- so every rank issues the same all-reduces in the same order whatever subset of its parameters took part in
    the step (a data-dependent branch that leaves a parameter without a gradient on one rank only delays that bucket to finish(), it
    does not reorder it: mismatched collective order hangs RCCL).  finish() issues what is left, waits, and the result is the MEAN over
    the ranks.  ONE backward per finish(): a second backward before finish() would add into gradients that are already being reduced, so
    its hooks raise.  Without an initialised process group everything degrades to a no-op average over one rank.
    `fp16_compression`: the wire carries fp16 of grad / world (divide BEFORE the reduce, like torch's fp16_compress_hook, so the sum of
    the ranks cannot overflow where the mean would not); arrival is converted back to fp32."""

    def __init__(self, params, bucket_mb=64.0, fp16_compression=False, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.fp16 = bool(fp16_compression)
        self.buckets = []                   # [flat, [params], set of ids seen this step]
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
        self._issued = 0                    # buckets [0, _issued) have their collective in flight (or done) this step
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(bi)) for bi, (_, ps, _) in enumerate(self.buckets) for p in ps]

    def _close(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)        # the gradient accumulates straight into the bucket
            off += p.numel()
        self.buckets.append((flat, ps, set()))

    def _active(self):
        return dist.is_initialized() and not (dist.get_world_size(self.group) == 1 and dist.get_backend(self.group) != "nccl")

    def _make_hook(self, bi):
        def hook(p):
            flat, ps, seen = self.buckets[bi]
            if bi < self._issued or id(p) in seen:
                raise RuntimeError("GradientBuckets: a second backward reached a bucket before finish(); gradient accumulation over several "
                                   "backward passes needs finish() after the LAST one only -- build the object with the hooks removed "
                                   "(remove()) for the earlier passes")
            seen.add(id(p))
            while self._issued < len(self.buckets) and len(self.buckets[self._issued][2]) == len(self.buckets[self._issued][1]):
                self._launch(self._issued)
        return hook

    def _launch(self, bi):
        assert bi == self._issued
        self._issued += 1
        flat, ps, seen = self.buckets[bi]
        if not self._active():
            return
        for p in ps:                                             # a hook-less path may have replaced .grad: keep the views authoritative
            if p.grad is not None and p.grad.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr():
                raise RuntimeError("GradientBuckets: a parameter's .grad was re-allocated outside its bucket")
        world = dist.get_world_size(self.group)
        if self.fp16 and flat.dtype == torch.float32:
            wire = (flat / world).half()
            self._work.append((dist.all_reduce(wire, group=self.group, async_op=True), flat, wire, 1))
        else:
            self._work.append((dist.all_reduce(flat, group=self.group, async_op=True), flat, None, world))

    def finish(self):
        """issue what is left (in bucket order), wait for every bucket, average.  Returns the number of all-reduces that ran."""
        while self._issued < len(self.buckets):                  # a bucket some parameter left incomplete this step, and all behind it
            self._launch(self._issued)
        n = len(self._work)
        for work, flat, wire, div in self._work:
            work.wait()
            if wire is not None:
                flat.copy_(wire)
            if div > 1:
                flat.div_(div)
        self._work = []
        self._issued = 0
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
