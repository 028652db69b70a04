"""bf16 rounding points of the mixed-precision path, for the CPU oracles (TEST INFRASTRUCTURE).

BASELINE.json configs[4] runs the matrix-core convolutions on bf16 operands with fp32 accumulation.  To tell bf16
ROUNDING from kernel ERROR, the oracles take an `emu=Bf16Emu()` argument and then round, in whatever precision they
run (fp32 / fp64), at the places where the HIP path stores bf16 (rehrseg_amd/ops.py `mixed_precision`):

  act(x)     an activation handed to / produced by a matrix-core layer: the value is rounded to bf16 on the way
             forward and its gradient on the way back (activation gradients are bf16 tensors on the device)
  weight(w)  the bf16 copy of an fp32 master weight: rounded forward, gradient passed through unrounded (weight
             gradients are fp32 on the device)
  fwd(x)     value rounded, gradient passed through: a tensor stored as bf16 whose gradient is NOT formed branch by branch
  grad(x)    value untouched, the TOTAL gradient arriving at x rounded: the device forms the gradient of a conv output
             (dz) in fp32 from all its contributions -- the normalised / gated activation and the statistics path -- and
             stores it once as bf16 (instnorm_bwd_apply: one store; SEGating: the scaled gradient is stored, then the
             per-channel constant of the gate path is added in place: act() on the branch + grad() on the total)

Statistics (InstanceNorm mean / variance, SE pool) are formed from the UNROUNDED accumulators, as the conv epilogues do.
Layers that stay fp32 on the device (C_in <= 2, the 1x1x1 logits layer, losses) are not rounded.  The emulation has the
device's rounding POINTS, not its bit pattern: fp32 accumulation order differs, so an element within an accumulation
error of a bf16 rounding boundary can round the other way (one bf16 ulp on that element).
"""
import torch


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class Bf16Emu:
    def act(self, x):
        return _RoundBoth.apply(x)

    def fwd(self, x):
        return x + (x.detach().to(torch.bfloat16).to(x.dtype) - x.detach())

    def grad(self, x):
        return _RoundGrad.apply(x)

    def weight(self, w):
        return w + (w.detach().to(torch.bfloat16).to(w.dtype) - w.detach())   # rounded value, straight-through gradient
