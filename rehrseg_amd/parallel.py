"""Patch-parallel training: one process per GPU, every rank draws its own patches
(the reference samples independent random patches per sample, utils/train_set.py:
109-121, 341-355; nothing in the forward couples samples), gradients averaged by
ONE exchange per step over RCCL/xGMI (torch.distributed backend "nccl").

Gradients live in a single flat fp32 buffer (`.grad` of every parameter is a view
into it), laid out in REVERSE registration order: backward produces the decoder's
gradients first, so the front of the buffer completes first.  The buffer is cut into
a few large buckets (default 64 MiB: xGMI is point-to-point and per-link bound, so
few large messages beat many small ones); a bucket's all-reduce is launched from a
post-accumulate hook the moment its last gradient is complete, i.e. the exchange
of the decoder's gradients overlaps the encoder's backward.  No flatten/unflatten
copies, no per-parameter messages.

A gradient reaches its slot by one of three routes:
  direct   conv weights on the device: the weight-gradient kernel writes into the slot itself
           (rehrseg_amd.ops registers them; only for weights used once in a step)
  stolen   small parameters get `.grad = None` at zero_grad, autograd hands over a fresh tensor
           (no add kernel), the values are copied into the slot bucket-wise right before the exchange
  in place autograd accumulates into the view (weights used several times, or after a plain
           `optimizer.zero_grad(set_to_none=False)`)
and whatever `.grad` holds when a bucket is exchanged is reconciled with the slot (`_collect`): a tensor that
is not the view is copied in and the view restored, a missing gradient contributes zeros -- so a plain
`optimizer.zero_grad()` (set_to_none=True) between steps is safe too.

Weight gradients on a second HIP stream (`wgrad_stream=True`, default on the device).  In the backward of a layer
the weight gradient does not feed the chain to the layer below -- only the input gradient does.  The direct-route
weight-gradient kernels are therefore launched on a side stream (after an event that marks dz complete) while the main
stream goes on with the input gradient and the HBM-bound InstanceNorm / SEGating passes of the next layer down: the
matrix-core kernel of one stream fills the tail rounds and the memory-bound passes of the other (cfg-3: 48.2 -> 45.5 ms per step, cfg-2: 61.4 -> 60.3, cfg-4: 43.6 -> 40.0).  Nothing reads a side-stream
result before the join, which the first side launch of a backward pass queues as an end-of-backward callback of the
autograd engine (and `reduce_gradients()` repeats): the kernels write into the flat buffer (no autograd accumulation), bias gradients are handed over
to parameters whose `.grad` is None, and a bucket's all-reduce launched from the side stream first waits for the main
one.  Same kernels, same operands, same results.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class PatchParallel:
    def __init__(self, module: torch.nn.Module, bucket_mb: float = 64, process_group=None, overlap: bool = True,
                 force_overlap: bool = False, direct=None, wgrad_stream: bool = True):
        """force_overlap: take the hook-launched bucket path even with a single rank (tests: the exchange code
        then runs, over a world-1 group, exactly as it does on 8 GPUs).
        direct: the parameters whose gradient the weight-gradient kernels may write in place (default: every
        5-D device parameter, i.e. the conv weights)."""
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
        self.buckets = []  # [start, end]
        self._params_of_bucket = []
        self._view = {}
        off = 0
        # The LAST bucket is the one exchange nothing hides (its gradients complete at the very end of backward), so the
        # tail of the buffer -- the last `bucket_elems` elements -- is cut into buckets that halve in size down to 1/16 of
        # a bucket: the exposed message is a few MiB instead of up to a whole bucket (FLAVR at cfg-2: 79 + 69 + 32 MiB
        # became 79 + 56 + 21 + 14 + 5 + 4 + 2 MiB: layer3 / layer2 go out while layer1 and the stem are still being computed).
        tail_from = total - bucket_elems
        cap = bucket_elems
        for p in reversed(self.params):
            n = p.numel()
            self._view[id(p)] = p.grad = self.flat[off:off + n].view_as(p)
            if self.buckets and off >= tail_from and off - self.buckets[-1][0] >= cap // 2 and cap > bucket_elems // 16:
                cap //= 2                                       # inside the tail: every new bucket half the previous cap
            if not self.buckets or off - self.buckets[-1][0] >= cap:
                self.buckets.append([off, off + n])
                self._params_of_bucket.append([])
            self.buckets[-1][1] = off + n
            self._params_of_bucket[-1].append(p)
            self._bucket_of[id(p)] = len(self.buckets) - 1
            off += n
        # conv weights: let the weight-gradient kernels write into the flat buffer directly (rehrseg_amd.ops)
        from . import ops
        self._ops = ops
        self._direct = [p for p in self.params if p.dim() == 5 and p.is_cuda] if direct is None else list(direct)
        for p in self._direct:
            ops.register_direct_grad(p, self)
        # every other parameter (biases, norm / gate parameters: dozens of tiny tensors) gets `.grad = None` at
        # zero_grad, so autograd hands its gradient over without an add kernel each (0.7-1.0 ms of 10 us launches
        # per step); the values are copied into the flat buffer bucket-wise, right before a bucket's exchange
        direct_ids = {id(p) for p in self._direct}
        self._steal = [p for p in self.params if id(p) not in direct_ids]
        self.exchange = self.world > 1 or force_overlap
        if force_overlap and not dist.is_initialized():
            raise ValueError("force_overlap needs an initialised process group (world size 1 is fine)")
        self.overlap = overlap and self.exchange
        # second stream for the direct-route weight gradients (see the module docstring); device parameters only
        self._side_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._side = self._side_stream if wgrad_stream else None
        self._side_used = False
        self._join_queued = False
        self._main = None
        self._reset_step()
        if self.overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad_ready)
        if self.world > 1:
            for p in module.parameters():  # identical starting point on every rank
                dist.broadcast(p.data, src=0, group=process_group)
            for b in module.buffers():
                dist.broadcast(b.data, src=0, group=process_group)
            if dev.type == "cuda":      # (written through .data: no version counter moved)
                from . import hip_backend
                hip_backend.invalidate_weight_forms()

    def _reset_step(self, uses=True):
        """uses=False keeps the forward use counts: the reference's loops clear gradients AFTER the forward
        (`loss = ...; opt.zero_grad(); loss.backward()`, train_all.py:135-137, 553-555), and the counts recorded
        during that forward decide whether the weight-gradient kernels may write in place."""
        self._written = set()                                   # ids of parameters a kernel wrote directly
        if uses:
            self._uses = {}                                     # id -> forward uses since the last backward
            self._in_backward = False
        self._ready = [set() for _ in self.buckets]             # per bucket: ids whose gradient is complete
        self._launched = [False] * len(self.buckets)
        self._works = []

    def close(self):
        """Drop the registrations in rehrseg_amd.ops (also happens by itself when this object dies)."""
        self._ops.unregister_direct_grad(self)

    # ------------------------------------------------------------------ bucket exchange
    def _collect(self, i, on_side=None):
        """Reconcile bucket i's slots with what `.grad` holds; afterwards every `.grad` is its view again.
        on_side: the stream this runs on when it is not the stream the gradients were produced on."""
        src, dst, zero = [], [], []
        for p in self._params_of_bucket[i]:
            v = self._view[id(p)]
            if p.grad is None:
                if id(p) not in self._written:
                    zero.append(v)                              # no gradient this step: contributes zeros
            elif p.grad is not v:
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        if zero:
            torch._foreach_zero_(zero)
        if src:
            torch._foreach_copy_(dst, src)
            if on_side is not None:
                # autograd allocated these on the backward's stream and `p.grad = v` above dropped the last reference:
                # the allocator must not hand the blocks out again before the side stream's copy has read them
                for t in src:
                    t.record_stream(on_side)

    def wgrad_stream(self):
        """The side stream the direct-route weight-gradient kernels run on (None: everything on the current stream)."""
        return self._side

    def set_wgrad_stream(self, enabled: bool):
        """Switch the side stream on / off between steps (bench.py times the kernels one by one with it off: the
        duration of a kernel that shares the chip with another stream's kernel says nothing about the kernel)."""
        self._join_side()
        self._side = self._side_stream if enabled else None

    def note_side_launch(self, main_stream):
        """A kernel is about to be launched on the side stream (called from inside a backward function);
        `main_stream` is the stream the backward runs on.  The first launch of a backward pass queues the join as an
        end-of-backward callback of the autograd engine: whoever reads a gradient on the main stream after
        `backward()` returns -- `reduce_gradients()`, an optimizer, a test -- sees completed kernels."""
        self._side_used = True
        self._main = main_stream
        if not self._join_queued:
            self._join_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self):
        self._join_queued = False
        if self._side is not None and self._side_used and self._main is not None:
            self._main.wait_stream(self._side)
            # (buckets still to be launched from reduce_gradients() run on the main stream, which is now behind the
            # side stream; _side_used stays set so that an exchange launched later in this step keeps its ordering)

    def join_side(self):
        """Public form of the join: called before a gradient that a side-stream kernel may still be writing is touched
        on the main stream (a second backward pass accumulating into a directly written weight gradient)."""
        self._join_side()

    def _join_side(self):
        """Main stream waits for everything launched on the side stream (before gradients are read on the main one)."""
        if self._side is not None and self._side_used:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
            self._side_used = False

    def _launch(self, i):
        s, e = self.buckets[i]
        if self._side is not None and self._side_used:
            # Gradients of this bucket may come from BOTH streams (a parameter counts as ready when its kernel has been
            # LAUNCHED).  The exchange is issued from the side stream after that stream has caught up with the other
            # one: RCCL orders itself behind the issuing stream, the main stream is not held up.
            cur = torch.cuda.current_stream(self._side.device)
            other = self._main if cur == self._side else cur
            if other is not None:
                self._side.wait_stream(other)
            with torch.cuda.stream(self._side):
                self._collect(i, on_side=self._side)
                self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._collect(i)
            self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._launched[i] = True

    def _on_grad_ready(self, p):
        i = self._bucket_of[id(p)]
        self._ready[i].add(id(p))                               # idempotent: a parameter counts once
        if len(self._ready[i]) == len(self._params_of_bucket[i]) and not self._launched[i]:
            self._launch(i)

    # ---- direct gradient writes (see rehrseg_amd.ops._direct_grad)
    def note_use(self, p):
        if self._in_backward:                                   # first forward use after a backward: a new step
            self._uses.clear()
            self._in_backward = False
        self._uses[id(p)] = self._uses.get(id(p), 0) + 1

    def may_write(self, p):
        """A kernel may overwrite p.grad only if this is the weight's single use in the step and nothing
        has been written yet (otherwise autograd's accumulation is the only correct route)."""
        self._in_backward = True
        return self._uses.get(id(p), 0) == 1 and id(p) not in self._written

    def was_written(self, p):
        return id(p) in self._written

    def grad_written(self, p):
        """A kernel has overwritten p.grad in place for this step: same bookkeeping as the autograd hook."""
        self._written.add(id(p))
        if self.overlap:
            self._on_grad_ready(p)

    def zero_grad(self):
        """Keeps the .grad views of the conv weights, hands the small parameters' `.grad` to autograd."""
        self.flat.zero_()
        for p in self._direct:
            p.grad = self._view[id(p)]
        for p in self._steal:
            p.grad = None
        self._reset_step(uses=False)                            # may run between forward and backward

    def reduce_gradients(self):
        """Sum over ranks, divide by world size; returns after the exchange completed.
        Buckets whose hooks did not fire (parameters without a gradient this step, or
        overlap disabled) are exchanged here.  Ends the step's bookkeeping, so the next backward
        starts clean whether or not `zero_grad()` of this object is called in between."""
        self._join_side()
        if not self.exchange:
            # single rank: handed-over gradients stay where autograd put them (the optimizer reads p.grad)
            self._written.clear()
            self._uses.clear()
            self._in_backward = False
            return
        for i in range(len(self.buckets)):
            if not self._launched[i]:
                self._launch(i)
        for w in self._works:
            w.wait()
        if self.world > 1:
            self.flat.div_(self.world)
        self._reset_step()

    def grad_bytes(self):
        return self.flat.numel() * self.flat.element_size()
