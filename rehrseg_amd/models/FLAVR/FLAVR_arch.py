"""FLAVR 3-D U-Net (`UNet_3D_3D`) on the MI355X kernels.

Drop-in for the reference's models/FLAVR/FLAVR_arch.py:117-248: same constructor
arguments, same forward signature (including the reference's spelling of
`return_inetermediate_*`), same return values, same in-place mean subtraction of
the caller's tensor, same state-dict keys/shapes.  torch.nn modules only hold
parameters; the arithmetic is rehrseg_amd.ops (HIP kernels):

  decoder[0], decoder[3]   Conv3d -> SEGating -> LeakyReLU(0.2)            one fused node each
  decoder[1,2,4]           ConvTranspose3d (3,4,4)/(1,2,2) -> SEGating -> LeakyReLU(0.2),
                           reading [previous, skip] as a *virtual* concat (no cat tensor is built)
  feature_fuse             the reference unbinds depth into channels and runs a Conv2d with
                           64*n_inputs channels (:201,:145); here it is the same contraction done
                           as a (n_inputs,3,3) Conv3d straight on the NDHWC tensor: no 537 MB
                           transpose at 128^3
  outconv                  ReflectionPad2d(3) + 7x7 conv, output rows padded to one MFMA tile
The few-kilobyte tail (split / tanh / softmax mixing of the UASR head) stays in torch.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from . import resnet_3D
from .resnet_3D import _SEParams


def joinTensors(X1, X2, type="concat"):
    if type == "concat":
        return torch.cat([X1, X2], dim=1)
    if type == "add":
        return X1 + X2
    return X1


class Conv_2d(nn.Module):
    """Parameter holder with the reference's key layout (`conv.0.{weight,bias}`)."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, padding=0, bias=False, batchnorm=False):
        super().__init__()
        if batchnorm:
            raise NotImplementedError("batchnorm=False is the only configuration REHRSeg uses")
        # the reference passes `bias=nn.InstanceNorm2d` (a truthy class) => bias=True, no norm
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size, stride, padding, bias=bool(bias)))
        self.kernel_size, self.padding = kernel_size, padding

    def as3d(self, n_inputs=None):
        """(weight, bias) as a 5-D kernel.  With n_inputs the 2-D input channels are
        depth-major blocks of 64 (cat(unbind(x, 2), 1)), i.e. a (n_inputs,k,k) Conv3d."""
        c = self.conv[0]
        w = c.weight
        co, ci, kh, kw = w.shape
        if n_inputs is None:
            return w.reshape(co, ci, 1, kh, kw), c.bias
        return w.reshape(co, n_inputs, ci // n_inputs, kh, kw).permute(0, 2, 1, 3, 4).contiguous(), c.bias


class upConv3D(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride, padding, upmode="transpose", batchnorm=False):
        super().__init__()
        if upmode != "transpose" or batchnorm:
            raise NotImplementedError("REHRSeg uses upmode='transpose', batchnorm=False (train_all.py:341-343)")
        self.stride, self.padding = stride, padding
        self.upconv = nn.Sequential(nn.ConvTranspose3d(in_ch, out_ch, kernel_size, stride, padding),
                                    _SEParams(out_ch))

    def forward(self, x, skip=None, act=ops.ACT_NONE, slope=0.0):
        c = self.upconv[0]
        return ops.fused_conv3d(x, c.weight, c.bias, self.stride, self.padding, x2=skip, transposed=True,
                                se=self.upconv[1].pair(), act=act, slope=slope)


class Conv_3d(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride=1, padding=0, bias=True, batchnorm=False):
        super().__init__()
        if batchnorm:
            raise NotImplementedError("batchnorm=False is the only configuration REHRSeg uses")
        self.stride, self.padding = stride, padding
        self.conv = nn.Sequential(nn.Conv3d(in_ch, out_ch, kernel_size, stride, padding, bias=bias),
                                  _SEParams(out_ch))

    def forward(self, x, skip=None, act=ops.ACT_NONE, slope=0.0):
        c = self.conv[0]
        return ops.fused_conv3d(x, c.weight, c.bias, self.stride, self.padding, x2=skip,
                                se=self.conv[1].pair(), act=act, slope=slope)


class UNet_3D_3D(nn.Module):
    def __init__(self, img_channels, block, n_inputs, n_outputs, batchnorm=False, joinType="concat",
                 upmode="transpose", use_uncertainty=False):
        super().__init__()
        if joinType != "concat":
            raise NotImplementedError("REHRSeg uses joinType='concat' (train_all.py:342)")
        nf = [512, 256, 128, 64]
        self.out_channels = img_channels * n_outputs
        self.joinType = joinType
        self.n_inputs = n_inputs
        self.n_outputs = n_outputs
        self.img_channels = img_channels
        self.use_uncertainty = use_uncertainty
        growth = 2
        if n_outputs > 1:
            resnet_3D.useBias = True
        self.encoder = getattr(resnet_3D, block)(pretrained=False, bn=batchnorm, img_channels=img_channels)
        self.decoder = nn.Sequential(
            Conv_3d(nf[0], nf[1], kernel_size=3, padding=1, bias=True, batchnorm=batchnorm),
            upConv3D(nf[1] * growth, nf[2], (3, 4, 4), (1, 2, 2), (1, 1, 1), upmode, batchnorm),
            upConv3D(nf[2] * growth, nf[3], (3, 4, 4), (1, 2, 2), (1, 1, 1), upmode, batchnorm),
            Conv_3d(nf[3] * growth, nf[3], kernel_size=3, padding=1, bias=True, batchnorm=batchnorm),
            upConv3D(nf[3] * growth, nf[3], (3, 4, 4), (1, 2, 2), (1, 1, 1), upmode, batchnorm),
        )
        self.feature_fuse = Conv_2d(nf[3] * n_inputs, nf[3] * n_inputs if use_uncertainty else nf[3], 3, 1, 1,
                                    batchnorm=batchnorm, bias=True)
        self.feature_fuse1 = Conv_2d(nf[3] * n_inputs, nf[3] * img_channels, 1, 1, batchnorm=batchnorm, bias=True)
        if use_uncertainty:
            self.uncertainty_early = Conv_2d(nf[3] * n_inputs, nf[3], 1, 1, batchnorm=batchnorm, bias=True)
            self.uncertainty_out = nn.Conv3d(nf[3] // n_outputs, 1, kernel_size=1, stride=1)
        self.outconv = nn.Sequential(nn.ReflectionPad2d(3),
                                     nn.Conv2d(nf[3], self.out_channels, kernel_size=7, stride=1, padding=0))

    def calc_out_patch_size(self, input_patch_size):
        x = torch.rand(tuple([1, self.img_channels] + list(input_patch_size))).float()
        x = x.to(next(self.parameters()).device)
        with torch.no_grad():
            out = self(x)
        if self.use_uncertainty:
            out = out[0]
        patch_size = list(out.shape[2:])
        patch_size[0] *= self.n_inputs
        return patch_size

    def _outconv(self, fused):
        """ReflectionPad2d(3) + Conv2d 7x7 on (N,64,1,H,W); rows padded to a 32-wide MFMA tile."""
        c = self.outconv[1]
        co = c.weight.shape[0]
        pad_rows = (-co) % 32
        w = F.pad(c.weight, (0, 0, 0, 0, 0, 0, 0, pad_rows)).unsqueeze(2)
        b = F.pad(c.bias, (0, pad_rows))
        x = F.pad(fused[:, :, 0], (3, 3, 3, 3), mode="reflect").unsqueeze(2)
        return ops.fused_conv3d(x, w, b, 1, 0)[:, :co, 0]

    def forward(self, images, return_inetermediate_uncertainty=False, return_inetermediate_feature=False):
        L = ops.ACT_LRELU
        if images.shape[2] != self.n_inputs and not return_inetermediate_feature:
            raise ValueError(f"depth {images.shape[2]} != n_inputs {self.n_inputs} (the reference's feature_fuse "
                             "requires them to be equal)")
        mean_ = images[:, 0:1].mean(2, keepdim=True).mean(3, keepdim=True).mean(4, keepdim=True)
        images[:, 0:1] = images[:, 0:1] - mean_  # in place on the caller's tensor, like the reference (:181)

        x_0, x_1, x_2, x_3, x_4 = self.encoder(images)
        if return_inetermediate_feature:
            return x_0, x_1, x_2, x_3, x_4

        dx_3 = self.decoder[0](x_4, None, L, 0.2)
        dx_2 = self.decoder[1](dx_3, x_3, L, 0.2)
        dx_1 = self.decoder[2](dx_2, x_2, L, 0.2)
        dx_0 = self.decoder[3](dx_1, x_1, L, 0.2)
        dx_out = self.decoder[4](dx_0, x_0, L, 0.2)

        wf, bf = self.feature_fuse.as3d(self.n_inputs)
        fused = ops.fused_conv3d(dx_out, wf, bf, 1, (0, 1, 1), act=L, slope=0.2)  # (N, C, 1, H, W)

        if self.use_uncertainty:
            w1, b1 = self.feature_fuse1.as3d()
            out = ops.fused_conv3d(fused, w1, b1, 1, 0)
            we, be = self.uncertainty_early.as3d()
            ue = ops.fused_conv3d(fused, we, be, 1, 0)
            if not return_inetermediate_uncertainty and ops.uasr_mix_supported(out, ue, self.n_outputs):
                # training / inference path: the candidate loop below as one pass (rehr_uasr_mix_*)
                return ops.uasr_mix(out, ue, self.uncertainty_out.weight, self.uncertainty_out.bias, self.n_outputs)
            out, ue = out[:, :, 0], ue[:, :, 0]
            out = torch.stack(torch.split(out, out.shape[1] // self.n_outputs, dim=1), dim=2)
            ue = torch.stack(torch.split(ue, ue.shape[1] // self.n_outputs, dim=1), dim=2)
            sm = torch.softmax(ue, dim=1)
            out_multi, out = out, 0
            imgs, uncs, segs = [], [], []
            for i in range(sm.shape[1]):
                img = (torch.tanh(out_multi[:, 2 * i:2 * i + 1]) + 1) / 2
                if return_inetermediate_uncertainty:
                    imgs.append(img)
                    uncs.append(sm[:, i:i + 1])
                    segs.append(out_multi[:, 2 * i + 1:2 * i + 2])
                out = out + torch.cat([img * sm[:, i:i + 1], out_multi[:, 2 * i + 1:2 * i + 2] * sm[:, i:i + 1]], 1)
            if return_inetermediate_uncertainty:
                return imgs, uncs, segs
            # uncertainty_out is a 1x1x1 conv to ONE channel (ref :151,245): a weighted channel sum -- written as
            # such, no vendor convolution library behind it (MIOpen picks its naive kernels for this shape)
            wu = self.uncertainty_out.weight.view(1, -1, 1, 1, 1)
            unc = torch.sigmoid((sm * wu).sum(1, keepdim=True) + self.uncertainty_out.bias.view(1, 1, 1, 1, 1))
            return out, unc

        out = self._outconv(fused)
        outs = torch.split(out, self.img_channels, dim=1)
        m2 = mean_.squeeze(2)
        if self.img_channels > 1:
            outs = [torch.cat([torch.tanh(o[:, 0:1] + m2), o[:, 1:2]], dim=1) for o in outs]
        else:
            outs = [o + m2 for o in outs]
        out = torch.stack(outs, dim=2)
        if return_inetermediate_uncertainty:
            return [], [], []  # the reference returns three empty-able lists here (:241-242)
        return out
