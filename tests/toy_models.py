"""Tiny deterministic stand-ins for a network, shared by the fixture generator
(tools/gen_golden_inference.py) and the tests of the whole-volume inference helpers: they only
need something position- and orientation-sensitive behind `model(x)`."""
import torch


class ToySegNet(torch.nn.Module):
    """(logits_lr (B,2,D,H,W), logits_hr (B,2,D*sep,H,W)); not flip-equivariant on purpose."""

    def __init__(self, sep=2):
        super().__init__()
        self.sep = sep

    def forward(self, x):
        B, _, D, H, W = x.shape
        ramp = (torch.arange(D, dtype=x.dtype).view(1, 1, D, 1, 1) * 0.25 +
                torch.arange(H, dtype=x.dtype).view(1, 1, 1, H, 1) * 0.0625 -
                torch.arange(W, dtype=x.dtype).view(1, 1, 1, 1, W) * 0.03125).to(x.device)
        sh = torch.roll(x, shifts=(1, 2), dims=(3, 4))
        lr = torch.cat([x * 0.5 + sh * 0.25 + ramp, -x * 0.75 + sh * sh * 0.125 - ramp * 0.5], 1)
        return lr, torch.repeat_interleave(lr, self.sep, dim=2) * 1.5
