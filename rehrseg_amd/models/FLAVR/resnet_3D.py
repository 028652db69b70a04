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


def encoder_on_windows(enc, xpad, nwin, upto):
    """x_0..x_upto of the `nwin` overlapping 4-slice windows xpad[:, :, w:w+4] of a depth-padded volume
    (B, C, nwin+3, H, W), each window with UNet_3D_3D.forward's mean subtraction on channel 0
    (FLAVR_arch.py:181) -- what `encoder(windows)` returns for get_intermediate_features' window batch
    (train_all.py:85-112), without convolving the shared slices once per window: the stem is linear, so the
    (7,7) part of each of its 3 depth taps runs once per volume slice (plus once on a constant-1 image for the
    mean term) and one HBM-bound pass assembles the windows.  No autograd (frozen teacher)."""
    s = enc.stem[0]
    B, C, Dp, H, W = xpad.shape
    if Dp != nwin + 3 or tuple(s.weight.shape[2:]) != (3, 7, 7):
        raise ValueError("encoder_on_windows: depth-padded volume of nwin + 3 slices, (3,7,7) stem")
    with torch.no_grad():
        sl = xpad.permute(0, 2, 1, 3, 4).reshape(B * Dp, C, 1, H, W)
        one = torch.zeros((1, C, 1, H, W), dtype=xpad.dtype, device=xpad.device)
        one[:, 0] = 1.0
        inp = ops.to_cl(torch.cat([sl, one], dim=0))
        g = [ops.fused_conv3d(inp, s.weight[:, :, kd:kd + 1].contiguous(), None, (1, 2, 2), (0, 3, 3), y_fp32=True)
             for kd in range(3)]
        ssum = xpad[:, 0].sum(dim=(2, 3))                                            # (B, Dp)
        mean = (ssum[:, 0:nwin] + ssum[:, 1:nwin + 1] + ssum[:, 2:nwin + 2] + ssum[:, 3:nwin + 3]) / (4.0 * H * W)
        okw = {"out_dtype": torch.bfloat16} if ops.is_mixed_precision() else {}   # layer1 takes bf16 then
        x0 = ops.get_backend().window_stem_assemble(g, mean.contiguous(), s.bias, B, nwin, ops.ACT_RELU, 0.0, **okw)
        feats = [x0]
        for i in range(1, upto + 1):
            feats.append(getattr(enc, f"layer{i}")(feats[-1]))
    return tuple(feats)


def unet_18(pretrained=False, bn=False, progress=True, img_channels=3, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained download is unavailable offline (the reference never enables it)")
    if bn:
        raise NotImplementedError("REHRSeg builds the encoder with batchnorm=False (train_all.py:341)")
    return _Encoder(img_channels)
