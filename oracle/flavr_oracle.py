"""Functional CPU restatement of the FLAVR 3D U-Net (TEST INFRASTRUCTURE).

Follows, operation by operation, the reference's
  models/FLAVR/FLAVR_arch.py:117-248  (UNet_3D_3D: ctor wiring and forward)
  models/FLAVR/resnet_3D.py:42-50     (BasicStem), :100-116 (SEGating),
  models/FLAVR/resnet_3D.py:118-151   (BasicBlock), :153-210 (VideoResNet layers)
written against a plain state-dict with torch.nn.functional so that it shares no
module code with either the reference or the HIP product.  Pinned by
tests/golden/flavr_*.npz, which tools/gen_golden.py produced by running the
reference itself in the build container.
"""
import torch
import torch.nn.functional as F

NF = (512, 256, 128, 64)  # FLAVR_arch.py:121


def se_gating(x, w, b):
    """resnet_3D.py:112-116: x * sigmoid(conv1x1x1(global_avg_pool(x)))."""
    pooled = x.mean(dim=(2, 3, 4), keepdim=True)
    return x * torch.sigmoid(F.conv3d(pooled, w, b))


def _block_emu(sd, p, x, stride, emu):
    """_block with the mixed-precision path's bf16 rounding points (oracle/bf16_emul.py): bf16 operands, SE pool
    from the unrounded conv output, conv2's output and the block output stored as bf16."""
    xq = emu.act(x)
    out = emu.act(torch.relu(F.conv3d(xq, emu.weight(sd[p + "conv1.0.weight"]), sd.get(p + "conv1.0.bias"), stride, 1)))
    y0 = emu.grad(F.conv3d(out, emu.weight(sd[p + "conv2.0.weight"]), sd.get(p + "conv2.0.bias"), 1, 1))
    gate = torch.sigmoid(F.conv3d(y0.mean(dim=(2, 3, 4), keepdim=True), sd[p + "fg.attn_layer.0.weight"],
                                  sd[p + "fg.attn_layer.0.bias"]))
    res = xq
    if (p + "downsample.0.weight") in sd:
        res = emu.act(F.conv3d(xq, emu.weight(sd[p + "downsample.0.weight"]), None, stride))
    return emu.act(torch.relu(emu.act(y0) * gate + res))


def _block(sd, p, x, stride, emu=None):
    """resnet_3D.py:140-151; conv bias only exists when useBias was set (:8,:33)."""
    if emu is not None:
        return _block_emu(sd, p, x, stride, emu)
    out = torch.relu(F.conv3d(x, sd[p + "conv1.0.weight"], sd.get(p + "conv1.0.bias"), stride, 1))
    out = F.conv3d(out, sd[p + "conv2.0.weight"], sd.get(p + "conv2.0.bias"), 1, 1)
    out = se_gating(out, sd[p + "fg.attn_layer.0.weight"], sd[p + "fg.attn_layer.0.bias"])
    res = x
    if (p + "downsample.0.weight") in sd:  # resnet_3D.py:196-200 (never has a bias)
        res = F.conv3d(x, sd[p + "downsample.0.weight"], None, stride)
    return torch.relu(out + res)


def encoder(sd, x, prefix="encoder.", emu=None, upto=4):
    """resnet_3D.py:183-189 with unet_18's layout (:238-261): strides (1,1,1),
    (1,2,2), (1,2,2), (1,1,1); depth is never reduced."""
    x0 = torch.relu(F.conv3d(x, sd[prefix + "stem.0.weight"], sd.get(prefix + "stem.0.bias"), (1, 2, 2), (1, 3, 3)))
    if emu is not None:
        x0 = emu.act(x0)     # the stem computes in fp32 on the fp32 image and stores bf16 (mixed precision)
    feats = [x0]
    cur = x0
    for li, stride in ((1, (1, 1, 1)), (2, (1, 2, 2)), (3, (1, 2, 2)), (4, (1, 1, 1)))[:upto]:
        cur = _block(sd, f"{prefix}layer{li}.0.", cur, stride, emu)
        cur = _block(sd, f"{prefix}layer{li}.1.", cur, (1, 1, 1), emu)
        feats.append(cur)
    return tuple(feats)


def _se_emu(y0, w, b, emu):
    """SEGating + LeakyReLU(0.2) behind a decoder conv on the mixed-precision path: gate from the unrounded conv
    output, the conv output and the block output stored as bf16 (the caller's LeakyReLU is fused on the device)."""
    y0 = emu.grad(y0)
    gate = torch.sigmoid(F.conv3d(y0.mean(dim=(2, 3, 4), keepdim=True), w, b))
    return emu.act(F.leaky_relu(emu.act(y0) * gate, 0.2))


def _dec_conv(sd, i, x, emu=None):
    """FLAVR_arch.py:72-88 Conv_3d = Conv3d(k3,p1,bias) -> SEGating.  (emu: LeakyReLU(0.2) included, see _se_emu)"""
    p = f"decoder.{i}.conv."
    if emu is not None:
        y = F.conv3d(emu.act(x), emu.weight(sd[p + "0.weight"]), sd[p + "0.bias"], 1, 1)
        return _se_emu(y, sd[p + "1.attn_layer.0.weight"], sd[p + "1.attn_layer.0.bias"], emu)
    y = F.conv3d(x, sd[p + "0.weight"], sd[p + "0.bias"], 1, 1)
    return se_gating(y, sd[p + "1.attn_layer.0.weight"], sd[p + "1.attn_layer.0.bias"])


def _dec_up(sd, i, x, emu=None):
    """FLAVR_arch.py:40-70 upConv3D(transpose) = ConvTranspose3d((3,4,4),(1,2,2),(1,1,1)) -> SEGating."""
    p = f"decoder.{i}.upconv."
    if emu is not None:
        y = F.conv_transpose3d(emu.act(x), emu.weight(sd[p + "0.weight"]), sd[p + "0.bias"], (1, 2, 2), (1, 1, 1))
        return _se_emu(y, sd[p + "1.attn_layer.0.weight"], sd[p + "1.attn_layer.0.bias"], emu)
    y = F.conv_transpose3d(x, sd[p + "0.weight"], sd[p + "0.bias"], (1, 2, 2), (1, 1, 1))
    return se_gating(y, sd[p + "1.attn_layer.0.weight"], sd[p + "1.attn_layer.0.bias"])


def uasr_head(out, ue, w_unc, b_unc, n_outputs):
    """FLAVR_arch.py:205-227, :244-246 from the two 1x1 responses (N, C, H, W) on the fused slice: candidate loop over
    the softmax weights, then uncertainty_out + sigmoid."""
    out = torch.stack(torch.split(out, out.shape[1] // n_outputs, dim=1), dim=2)
    ue = torch.stack(torch.split(ue, ue.shape[1] // n_outputs, dim=1), dim=2)
    sm = torch.softmax(ue, dim=1)
    total = 0
    for i in range(sm.shape[1]):
        img = (torch.tanh(out[:, 2 * i:2 * i + 1]) + 1) / 2 * sm[:, i:i + 1]
        seg = out[:, 2 * i + 1:2 * i + 2] * sm[:, i:i + 1]
        total = total + torch.cat([img, seg], dim=1)
    return total, torch.sigmoid(F.conv3d(sm, w_unc, b_unc))


def _unet_tail_emu(sd, x0, x1, x2, x3, x4, mean_, img_channels, emu):
    """Decoder and tail of unet_3d_3d (no uncertainty head) with the mixed-precision path's rounding points."""
    d3 = torch.cat([_dec_conv(sd, 0, x4, emu), x3], 1)
    d2 = torch.cat([_dec_up(sd, 1, d3, emu), x2], 1)
    d1 = torch.cat([_dec_up(sd, 2, d2, emu), x1], 1)
    d0 = torch.cat([_dec_conv(sd, 3, d1, emu), x0], 1)
    dout = _dec_up(sd, 4, d0, emu)
    dout = torch.cat(torch.unbind(dout, 2), 1)
    out = emu.act(F.leaky_relu(F.conv2d(emu.act(dout), emu.weight(sd["feature_fuse.conv.0.weight"]),
                                        sd["feature_fuse.conv.0.bias"], 1, 1), 0.2))
    out = emu.act(F.conv2d(F.pad(out, (3, 3, 3, 3), mode="reflect"), emu.weight(sd["outconv.1.weight"]),
                           sd["outconv.1.bias"]))
    outs = torch.split(out, img_channels, dim=1)
    m2 = mean_.squeeze(2)
    if img_channels > 1:
        outs = [torch.cat([torch.tanh(o[:, 0:1] + m2), o[:, 1:2]], dim=1) for o in outs]
    else:
        outs = [o + m2 for o in outs]
    return torch.stack(outs, dim=2)


def unet_3d_3d(sd, images, img_channels, n_inputs, n_outputs, use_uncertainty=False,
               return_intermediate_feature=False, emu=None, upto=4):
    """FLAVR_arch.py:169-248.  NOTE: like the reference (:180-181) this subtracts
    the mean of channel 0 from `images` IN PLACE.
    emu: oracle/bf16_emul.Bf16Emu() rounds where the mixed-precision device path stores bf16 (None: plain)."""
    lrelu = lambda t: F.leaky_relu(t, 0.2)
    mean_ = images[:, 0:1].mean(2, keepdim=True).mean(3, keepdim=True).mean(4, keepdim=True)
    images[:, 0:1] = images[:, 0:1] - mean_
    if return_intermediate_feature and upto < 4:
        return encoder(sd, images, emu=emu, upto=upto)
    x0, x1, x2, x3, x4 = encoder(sd, images, emu=emu)
    if return_intermediate_feature:
        return x0, x1, x2, x3, x4
    if emu is not None:
        if use_uncertainty:
            raise NotImplementedError("bf16 emulation of the UASR head")
        return _unet_tail_emu(sd, x0, x1, x2, x3, x4, mean_, img_channels, emu)
    d3 = torch.cat([lrelu(_dec_conv(sd, 0, x4)), x3], 1)
    d2 = torch.cat([lrelu(_dec_up(sd, 1, d3)), x2], 1)
    d1 = torch.cat([lrelu(_dec_up(sd, 2, d2)), x1], 1)
    d0 = torch.cat([lrelu(_dec_conv(sd, 3, d1)), x0], 1)
    dout = lrelu(_dec_up(sd, 4, d0))
    dout = torch.cat(torch.unbind(dout, 2), 1)  # depth -> channels (:201)

    if use_uncertainty:  # :203-227, :244-246
        dout = lrelu(F.conv2d(dout, sd["feature_fuse.conv.0.weight"], sd["feature_fuse.conv.0.bias"], 1, 1))
        out = F.conv2d(dout, sd["feature_fuse1.conv.0.weight"], sd["feature_fuse1.conv.0.bias"])
        ue = F.conv2d(dout, sd["uncertainty_early.conv.0.weight"], sd["uncertainty_early.conv.0.bias"])
        return uasr_head(out, ue, sd["uncertainty_out.weight"], sd["uncertainty_out.bias"], n_outputs)

    out = lrelu(F.conv2d(dout, sd["feature_fuse.conv.0.weight"], sd["feature_fuse.conv.0.bias"], 1, 1))
    out = F.conv2d(F.pad(out, (3, 3, 3, 3), mode="reflect"), sd["outconv.1.weight"], sd["outconv.1.bias"])
    outs = torch.split(out, img_channels, dim=1)
    m2 = mean_.squeeze(2)
    if img_channels > 1:
        outs = [torch.cat([torch.tanh(o[:, 0:1] + m2), o[:, 1:2]], dim=1) for o in outs]
    else:
        outs = [o + m2 for o in outs]
    return torch.stack(outs, dim=2)


def flavr_shapes(img_channels, n_inputs, n_outputs, use_uncertainty=False, enc_bias=None):
    """Parameter name -> shape of UNet_3D_3D(img_channels,'unet_18',n_inputs,n_outputs)
    (FLAVR_arch.py:118-156; encoder convs carry a bias iff n_outputs > 1, :133-134)."""
    if enc_bias is None:
        enc_bias = n_outputs > 1
    s = {}

    def conv(name, co, ci, k, bias):
        s[name + ".weight"] = (co, ci) + tuple(k)
        if bias:
            s[name + ".bias"] = (co,)

    conv("encoder.stem.0", 64, img_channels, (3, 7, 7), enc_bias)
    inp = 64
    for li, planes in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in (0, 1):
            p = f"encoder.layer{li}.{bi}."
            conv(p + "conv1.0", planes, inp if bi == 0 else planes, (3, 3, 3), enc_bias)
            conv(p + "conv2.0", planes, planes, (3, 3, 3), enc_bias)
            conv(p + "fg.attn_layer.0", planes, planes, (1, 1, 1), True)
            if bi == 0 and li > 1:
                conv(p + "downsample.0", planes, inp, (1, 1, 1), False)
        inp = planes
    conv("decoder.0.conv.0", 256, 512, (3, 3, 3), True)
    conv("decoder.0.conv.1.attn_layer.0", 256, 256, (1, 1, 1), True)
    for i, (ci, co) in ((1, (512, 128)), (2, (256, 64)), (4, (128, 64))):
        s[f"decoder.{i}.upconv.0.weight"] = (ci, co, 3, 4, 4)
        s[f"decoder.{i}.upconv.0.bias"] = (co,)
        conv(f"decoder.{i}.upconv.1.attn_layer.0", co, co, (1, 1, 1), True)
    conv("decoder.3.conv.0", 64, 128, (3, 3, 3), True)
    conv("decoder.3.conv.1.attn_layer.0", 64, 64, (1, 1, 1), True)
    conv("feature_fuse.conv.0", 64 * n_inputs if use_uncertainty else 64, 64 * n_inputs, (3, 3), True)
    conv("feature_fuse1.conv.0", 64 * img_channels, 64 * n_inputs, (1, 1), True)
    if use_uncertainty:
        conv("uncertainty_early.conv.0", 64, 64 * n_inputs, (1, 1), True)
        conv("uncertainty_out", 1, 64 // n_outputs, (1, 1, 1), True)
    conv("outconv.1", img_channels * n_outputs, 64, (7, 7), True)
    return s
