"""R3D-18 encoder of the FLAVR U-Net on the MI355X kernels.

Drop-in for the reference's models/FLAVR/resnet_3D.py: same constructor
(`unet_18(pretrained, bn, img_channels)`), same forward contract (returns
x_0..x_4), same parameter names (`stem.0`, `layerL.B.conv{1,2}.0`,
`layerL.B.fg.attn_layer.0`, `layerL.0.downsample.0`) so reference checkpoints
load unchanged.  torch.nn modules are used only as parameter containers; every
forward/backward computation goes through rehrseg_amd.ops (HIP kernels):

  stem      Conv3d (3,7,7)/(1,2,2) + ReLU        -> thin-input direct kernel   (ref :42-50)
  block     conv1+ReLU, conv2 -> SEGating -> +res -> ReLU as two fused launches (ref :118-151)
  shortcut  1x1x1 strided projection              -> gather-GEMM               (ref :196-200)
"""
import torch
import torch.nn as nn

from ... import ops

__all__ = ["unet_18"]

# Process-global like the reference (:8): FLAVR_arch flips it when n_outputs > 1.
useBias = False


class _SEParams(nn.Module):
    """Parameter holder for SEGating (ref :100-116): attn_layer.0 = Conv3d(C, C, 1)."""

    def __init__(self, planes):
        super().__init__()
        self.attn_layer = nn.Sequential(nn.Conv3d(planes, planes, kernel_size=1, bias=True))

    def pair(self):
        c = self.attn_layer[0]
        return c.weight, c.bias


def _conv_holder(cin, cout, k, stride, pad, bias):
    return nn.Sequential(nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=bias))


class _Block(nn.Module):
    def __init__(self, inplanes, planes, stride, project):
        super().__init__()
        self.stride = stride
        self.conv1 = _conv_holder(inplanes, planes, 3, stride, 1, useBias)
        self.conv2 = _conv_holder(planes, planes, 3, 1, 1, useBias)
        self.fg = _SEParams(planes)
        self.downsample = _conv_holder(inplanes, planes, 1, stride, 0, False) if project else None

    def forward(self, x):
        c1, c2 = self.conv1[0], self.conv2[0]
        out = ops.fused_conv3d(x, c1.weight, c1.bias, self.stride, 1, act=ops.ACT_RELU)
        res = x
        if self.downsample is not None:
            res = ops.fused_conv3d(x, self.downsample[0].weight, None, self.stride, 0)
        return ops.fused_conv3d(out, c2.weight, c2.bias, 1, 1, se=self.fg.pair(), res=res, act=ops.ACT_RELU)


class _Encoder(nn.Module):
    def __init__(self, img_channels):
        super().__init__()
        if img_channels not in (1, 2):
            raise NotImplementedError("the thin-input stem kernel covers img_channels in {1, 2} (what REHRSeg uses)")
        self.stem = _conv_holder(img_channels, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), useBias)
        cfg = ((64, 64, (1, 1, 1), False), (64, 128, (1, 2, 2), True), (128, 256, (1, 2, 2), True),
               (256, 512, (1, 1, 1), True))
        for i, (cin, cout, stride, project) in enumerate(cfg, start=1):
            setattr(self, f"layer{i}", nn.Sequential(_Block(cin, cout, stride, project),
                                                      _Block(cout, cout, (1, 1, 1), False)))
        for m in self.modules():  # ref :212-218
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x, upto=4):
        """x_0..x_upto.  `upto` < 4 stops early: the distillation teacher only needs x_1
        (train_all.py:550), which removes ~84 % of the encoder's FLOPs with identical results."""
        s = self.stem[0]
        feats = [ops.fused_conv3d(x, s.weight, s.bias, (1, 2, 2), (1, 3, 3), act=ops.ACT_RELU)]
        for i in range(1, upto + 1):
            feats.append(getattr(self, f"layer{i}")(feats[-1]))
        return tuple(feats)


def unet_18(pretrained=False, bn=False, progress=True, img_channels=3, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained download is unavailable offline (the reference never enables it)")
    if bn:
        raise NotImplementedError("REHRSeg builds the encoder with batchnorm=False (train_all.py:341)")
    return _Encoder(img_channels)
