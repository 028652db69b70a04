"""Patch-parallel training: one process per GPU, every rank draws its own patches
(the reference samples independent random patches per sample, utils/train_set.py:
109-121, 341-355; nothing in the forward couples samples), gradients averaged by
ONE exchange per step over RCCL/xGMI (torch.distributed backend "nccl").

Gradients live in a single flat fp32 buffer (`.grad` of every parameter is a view
into it), laid out in REVERSE registration order: backward produces the decoder's
gradients first, so the front of the buffer completes first.  The buffer is cut into
a few large buckets (default 64 MiB: xGMI is point-to-point and per-link bound, so
few large messages beat many small ones); a bucket's all-reduce is launched from a
post-accumulate hook the moment its last gradient has been written, i.e. the exchange
of the decoder's gradients overlaps the encoder's backward.  No flatten/unflatten
copies, no per-parameter messages.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class PatchParallel:
    def __init__(self, module: torch.nn.Module, bucket_mb: float = 64, process_group=None, overlap: bool = True):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        bucket_elems = max(1, int(bucket_mb * (1 << 20)) // self.flat.element_size())
        # reverse registration order ~ the order in which backward finishes the gradients
        self._bucket_of = {}
        self.buckets = []  # [start, end, n_params]
        off = 0
        for p in reversed(self.params):
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            if not self.buckets or off - self.buckets[-1][0] >= bucket_elems:
                self.buckets.append([off, off + n, 0])
            b = self.buckets[-1]
            b[1] = off + n
            b[2] += 1
            self._bucket_of[p] = len(self.buckets) - 1
            off += n
        # conv weights: let the weight-gradient kernels write into the flat buffer directly (rehrseg_amd.ops)
        from . import ops
        self._written = set()
        for p in self.params:
            if p.dim() == 5 and p.is_cuda:
                ops._direct_grad[p.data_ptr()] = (p, self)
        # every other parameter (biases, norm / gate parameters: dozens of tiny tensors) gets `.grad = None` at
        # zero_grad, so autograd hands its gradient over without an add kernel each (0.7-1.0 ms of 10 us launches
        # per step); the values are copied into the flat buffer bucket-wise, right before a bucket's exchange
        self._steal = [p for p in self.params if not (p.dim() == 5 and p.is_cuda)]
        self._view = {id(p): p.grad for p in self._steal}
        self._steal_of_bucket = [[] for _ in self.buckets]
        for p in self._steal:
            self._steal_of_bucket[self._bucket_of[p]].append(p)
        self.overlap = overlap and self.world > 1
        self._pending = [b[2] for b in self.buckets]
        self._works = []
        self._launched = [False] * len(self.buckets)
        if self.overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad_ready)
        if self.world > 1:
            for p in module.parameters():  # identical starting point on every rank
                dist.broadcast(p.data, src=0, group=process_group)
            for b in module.buffers():
                dist.broadcast(b.data, src=0, group=process_group)

    # ------------------------------------------------------------------ bucket exchange
    def _collect(self, i):
        """Bucket i's handed-over gradients -> their slots of the flat buffer; `.grad` becomes the view again."""
        ps = [p for p in self._steal_of_bucket[i] if p.grad is not None and p.grad is not self._view[id(p)]]
        if ps:
            torch._foreach_copy_([self._view[id(p)] for p in ps], [p.grad for p in ps])
            for p in ps:
                p.grad = self._view[id(p)]

    def _launch(self, i):
        s, e, _ = self.buckets[i]
        self._collect(i)
        self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._launched[i] = True

    def _on_grad_ready(self, p):
        i = self._bucket_of[p]
        self._pending[i] -= 1
        if self._pending[i] == 0 and not self._launched[i]:
            self._launch(i)

    # ---- direct gradient writes (see rehrseg_amd.ops._direct_grad)
    def was_written(self, p):
        return id(p) in self._written

    def grad_written(self, p):
        """A kernel has overwritten p.grad in place for this step: same bookkeeping as the autograd hook."""
        self._written.add(id(p))
        if self.overlap:
            self._on_grad_ready(p)

    def zero_grad(self):
        """Keeps the .grad views (optimizer.zero_grad(set_to_none=True) would drop them)."""
        self.flat.zero_()
        for p in self._steal:
            p.grad = None
        self._written.clear()
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []

    def reduce_gradients(self):
        """Sum over ranks, divide by world size; returns after the exchange completed.
        Buckets whose hooks did not fire (parameters without a gradient this step, or
        overlap disabled) are exchanged here."""
        if self.world == 1:
            return  # (handed-over gradients stay where autograd put them: the optimizer reads p.grad)
        for i in range(len(self.buckets)):
            if not self._launched[i]:
                self._launch(i)
        for w in self._works:
            w.wait()
        self._works = []
        self.flat.div_(self.world)
        for p in self._steal:  # parameters without a gradient this step: back to the (zero) view, like before
            if p.grad is None:
                p.grad = self._view[id(p)]

    def grad_bytes(self):
        return self.flat.numel() * self.flat.element_size()
