"""Patch-parallel training: one process per GPU, every rank draws its own patches
(the reference samples independent random patches per sample, utils/train_set.py:
109-121, 341-355; nothing in the forward couples samples), gradients averaged by
ONE exchange per step over RCCL/xGMI (torch.distributed backend "nccl").

Gradients live in a single flat fp32 buffer (`.grad` of every parameter is a view
into it), so the exchange is a handful of large all-reduces (default 64 MiB
buckets: xGMI is point-to-point, per-link bound, so few large messages beat many
small ones) with no flatten/unflatten copies.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class PatchParallel:
    def __init__(self, module: torch.nn.Module, bucket_mb: int = 64, process_group=None):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self.bucket_elems = max(1, (bucket_mb << 20) // self.flat.element_size())
        if self.world > 1:
            for p in module.parameters():  # identical starting point on every rank
                dist.broadcast(p.data, src=0, group=process_group)
            for b in module.buffers():
                dist.broadcast(b.data, src=0, group=process_group)

    def zero_grad(self):
        """Keeps the .grad views (optimizer.zero_grad(set_to_none=True) would drop them)."""
        self.flat.zero_()

    def reduce_gradients(self):
        """Sum over ranks, divide by world size; returns after the exchange completed."""
        if self.world == 1:
            return
        works = []
        for s in range(0, self.flat.numel(), self.bucket_elems):
            works.append(dist.all_reduce(self.flat[s:s + self.bucket_elems], op=dist.ReduceOp.SUM, group=self.group,
                                         async_op=True))
        for w in works:
            w.wait()
        self.flat.div_(self.world)

    def grad_bytes(self):
        return self.flat.numel() * self.flat.element_size()
