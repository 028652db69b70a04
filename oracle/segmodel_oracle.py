"""Functional CPU restatement of SegModel (TEST INFRASTRUCTURE) -- in-reference parts PINNED, bases UNPINNED.

In-reference parts followed: models/seg_model.py:26-58 (MyUnetDecoder.forward),
:174-210 (SegModel: sr_head, depth-only trilinear upsample align_corners=True).
The encoder/decoder bases come from dynamic_network_architectures==0.3.1
(requirements.txt:20; call sites models/seg_model.py:9-10,153,174-191,
train_all.py:474-493), which is absent offline and not vendored; its published
block semantics are restated: per stage n_conv x [Conv3d(k, pad=(k-1)//2, stride on
the first conv) -> InstanceNorm3d(eps, affine) -> LeakyReLU(0.01)], decoder stage =
ConvTranspose3d(kernel=stride) -> cat(skip) -> conv blocks, 1x1x1 seg layers.
The reference ships no test or fixture for that package boundary, so the BASES stay
"parity unpinned".  The parts the reference itself defines are pinned: tools/gen_golden_segmodel.py
imports the reference's models/seg_model.py over eager-torch stand-ins for the two absent modules and
records its SegModel.forward / MyUnetDecoder.forward outputs, losses and gradients
(tests/golden/segmodel_{small,aniso4}.npz); this oracle and the HIP SegModel are checked against them
(tests/test_segmodel_golden_cpu.py, tests/test_segmodel_golden_gpu.py).
"""
import torch
import torch.nn.functional as F


def _t(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * 3


def _cna(sd, p, x, k, stride, eps, slope, emu=None):
    k = _t(k)
    pad = tuple((i - 1) // 2 for i in k)
    if emu is None:
        y = F.conv3d(x, sd[p + "conv.weight"], sd.get(p + "conv.bias"), _t(stride), pad)
        y = F.instance_norm(y, weight=sd[p + "norm.weight"], bias=sd[p + "norm.bias"], eps=eps)
        return F.leaky_relu(y, slope)
    # mixed precision (oracle/bf16_emul.py): bf16 operands (thin-input layers: fp32 operands, the image is fp32),
    # statistics from the unrounded accumulators, the conv output and the block output stored as bf16
    if x.shape[1] <= 2:
        y = F.conv3d(x, sd[p + "conv.weight"], sd.get(p + "conv.bias"), _t(stride), pad)
    else:
        y = F.conv3d(emu.act(x), emu.weight(sd[p + "conv.weight"]), sd.get(p + "conv.bias"), _t(stride), pad)
    y = emu.grad(y)                                       # dz: all contributions summed in fp32, stored once as bf16
    mean = y.mean(dim=(2, 3, 4), keepdim=True)
    var = y.var(dim=(2, 3, 4), unbiased=False, keepdim=True)
    g, b = sd[p + "norm.weight"].view(1, -1, 1, 1, 1), sd[p + "norm.bias"].view(1, -1, 1, 1, 1)
    return emu.act(F.leaky_relu((emu.fwd(y) - mean) * torch.rsqrt(var + eps) * g + b, slope))


def _sr_head0_emu(sd, feats, upscale, emu):
    """sr_head.0 as the device computes it in mixed precision: the in-plane 3x3 part of each depth tap on the
    LOW-resolution features (48 bf16 response channels), then depth interpolation + depth-tap sum + bias + ReLU."""
    w, b = emu.weight(sd["sr_head.0.weight"]), sd["sr_head.0.bias"]
    x = emu.act(feats)
    tot = 0
    for kd in range(3):
        r = emu.act(F.conv3d(x, w[:, :, kd:kd + 1], None, 1, (0, 1, 1)))
        u = F.interpolate(r, scale_factor=(upscale, 1, 1), mode="trilinear", align_corners=True)
        u = F.pad(u, (0, 0, 0, 0, 1, 1))                                     # the 3-tap depth conv's zero padding
        tot = tot + u[:, :, kd:kd + u.shape[2] - 2]
    return emu.act(torch.relu(tot + b.view(1, -1, 1, 1, 1)))


def seg_model(sd, x, cfg, return_features=False, emu=None):
    """cfg: dict(n_stages, features_per_stage, kernel_sizes, strides, n_conv_per_stage,
    n_conv_per_stage_decoder, num_classes, upscale, eps, slope, deep_supervision).
    emu: oracle/bf16_emul.Bf16Emu() rounds where the mixed-precision device path stores bf16 (None: plain)."""
    eps, slope = cfg.get("eps", 1e-5), cfg.get("slope", 0.01)
    n = cfg["n_stages"]
    skips = []
    for s in range(n):
        for i in range(cfg["n_conv_per_stage"][s]):
            x = _cna(sd, f"encoder.stages.{s}.0.convs.{i}.", x, cfg["kernel_sizes"][s],
                     cfg["strides"][s] if i == 0 else 1, eps, slope, emu)
        skips.append(x)
    lres, segs, feats = skips[-1], [], None
    for s in range(n - 1):
        st = _t(cfg["strides"][-(s + 1)])
        if emu is None:
            up = F.conv_transpose3d(lres, sd[f"decoder.transpconvs.{s}.weight"], sd.get(f"decoder.transpconvs.{s}.bias"), st)
        else:
            up = emu.act(F.conv_transpose3d(emu.act(lres), emu.weight(sd[f"decoder.transpconvs.{s}.weight"]),
                                            sd.get(f"decoder.transpconvs.{s}.bias"), st))
        x = torch.cat((up, skips[-(s + 2)]), 1)
        for i in range(cfg["n_conv_per_stage_decoder"][s]):
            x = _cna(sd, f"decoder.stages.{s}.convs.{i}.", x, cfg["kernel_sizes"][-(s + 2)], 1, eps, slope, emu)
        if s == n - 2:
            feats = x
        xs = x if emu is None else emu.act(x)   # the fp32 logits layer reads the bf16 features; its input gradient is cast back
        if cfg.get("deep_supervision", False):
            segs.append(F.conv3d(xs, sd[f"decoder.seg_layers.{s}.weight"], sd[f"decoder.seg_layers.{s}.bias"]))
        elif s == n - 2:
            segs.append(F.conv3d(xs, sd[f"decoder.seg_layers.{n - 2}.weight"], sd[f"decoder.seg_layers.{n - 2}.bias"]))
        lres = x
    segs = segs[::-1]
    out = segs if cfg.get("deep_supervision", False) else segs[0]
    if emu is None:
        up = F.interpolate(feats, scale_factor=(cfg["upscale"], 1, 1), mode="trilinear", align_corners=True)
        up = torch.relu(F.conv3d(up, sd["sr_head.0.weight"], sd["sr_head.0.bias"], 1, 1))
        up = F.conv3d(up, sd["sr_head.2.weight"], sd["sr_head.2.bias"], 1, 2)
    else:
        up = _sr_head0_emu(sd, feats, cfg["upscale"], emu)
        up = F.conv3d(up, emu.weight(sd["sr_head.2.weight"]), sd["sr_head.2.bias"], 1, 2)   # bf16 operands, fp32 logits
    if return_features:
        return out, up, skips
    return out, up


def segmodel_shapes(cfg, input_channels=1):
    """Parameter name -> shape, primary keys only (no all_modules / decoder.encoder aliases)."""
    s = {}
    cin = input_channels
    n = cfg["n_stages"]
    f = cfg["features_per_stage"]
    for st in range(n):
        for i in range(cfg["n_conv_per_stage"][st]):
            p = f"encoder.stages.{st}.0.convs.{i}."
            s[p + "conv.weight"] = (f[st], cin if i == 0 else f[st]) + _t(cfg["kernel_sizes"][st])
            s[p + "conv.bias"] = (f[st],)
            s[p + "norm.weight"] = (f[st],)
            s[p + "norm.bias"] = (f[st],)
        cin = f[st]
    for st in range(n - 1):
        below, skip = f[-(st + 1)], f[-(st + 2)]
        s[f"decoder.transpconvs.{st}.weight"] = (below, skip) + _t(cfg["strides"][-(st + 1)])
        s[f"decoder.transpconvs.{st}.bias"] = (skip,)
        for i in range(cfg["n_conv_per_stage_decoder"][st]):
            p = f"decoder.stages.{st}.convs.{i}."
            s[p + "conv.weight"] = (skip, 2 * skip if i == 0 else skip) + _t(cfg["kernel_sizes"][-(st + 2)])
            s[p + "conv.bias"] = (skip,)
            s[p + "norm.weight"] = (skip,)
            s[p + "norm.bias"] = (skip,)
        s[f"decoder.seg_layers.{st}.weight"] = (cfg["num_classes"], skip, 1, 1, 1)
        s[f"decoder.seg_layers.{st}.bias"] = (cfg["num_classes"],)
    s["sr_head.0.weight"] = (16, 32, 3, 3, 3)
    s["sr_head.0.bias"] = (16,)
    s["sr_head.2.weight"] = (cfg["num_classes"], 16, 5, 5, 5)
    s["sr_head.2.bias"] = (cfg["num_classes"],)
    return s


ISO_PLAN = dict(n_stages=6, features_per_stage=[32, 64, 128, 256, 320, 320], kernel_sizes=[[3, 3, 3]] * 6,
                strides=[[1, 1, 1]] + [[2, 2, 2]] * 5, n_conv_per_stage=[2] * 6, n_conv_per_stage_decoder=[2] * 5,
                num_classes=2, upscale=4)
# distillation-compatible plan (stage-1 stride (1,2,2); DS scale table utils/seg_utils.py:364)
ANISO_PLAN = dict(n_stages=6, features_per_stage=[32, 64, 128, 256, 320, 320],
                  kernel_sizes=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
                  strides=[[1, 1, 1], [1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2]],
                  n_conv_per_stage=[2] * 6, n_conv_per_stage_decoder=[2] * 5, num_classes=2, upscale=4)
