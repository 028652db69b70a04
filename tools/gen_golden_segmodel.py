"""G8 fixtures: the REFERENCE's own `SegModel` / `MyUnetDecoder` and
`DC_and_weighted_CE_loss`, run over plain-torch stand-ins for the absent third-party bases
(build container only; see tools/gen_golden.py for the import recipe).

What is the reference's code in these fixtures (pinned): MyUnetDecoder.forward
(models/seg_model.py:26-58), SegModel.__init__/forward incl. the depth-only trilinear upsample
and sr_head (:153-210), RobustCrossEntropyLoss / DC_and_weighted_CE_loss.forward / _build_loss
(utils/seg_utils.py:289-372).  What is NOT (stays "parity unpinned"): the bases written below from
the published semantics of dynamic_network_architectures==0.3.1 (PlainConvUNet, PlainConvEncoder,
UNetDecoder.__init__, StackedConvBlocks, ConvDropoutNormReLU) and nnunetv2==2.3.1
(MemoryEfficientSoftDiceLoss, softmax_helper_dim1) -- both absent offline, neither vendored.
The stand-ins run torch.nn eager ops only and are registered in sys.modules under the packages'
module names before the reference is imported.

    python tools/gen_golden_segmodel.py      # rewrites tests/golden/segmodel_*.npz, seg_losses.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden import OUT, _Finder  # noqa: E402
from oracle.detinit import det_input, det_tensor  # noqa: E402


# ----------------------------------------------------------------------------- stand-in bases (eager torch)
def _l3(v):
    return list(v) if isinstance(v, (tuple, list)) else [v] * 3


class ConvDropoutNormReLU(nn.Module):
    def __init__(self, conv_op, input_channels, output_channels, kernel_size, stride, conv_bias=False, norm_op=None,
                 norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None, nonlin_kwargs=None,
                 nonlin_first=False):
        super().__init__()
        kernel_size, stride = _l3(kernel_size), _l3(stride)
        seq = []
        self.conv = conv_op(input_channels, output_channels, kernel_size, stride,
                            padding=[(k - 1) // 2 for k in kernel_size], dilation=1, bias=conv_bias)
        seq.append(self.conv)
        if dropout_op is not None:
            self.dropout = dropout_op(**dropout_op_kwargs)
            seq.append(self.dropout)
        if norm_op is not None:
            self.norm = norm_op(output_channels, **norm_op_kwargs)
            seq.append(self.norm)
        if nonlin is not None:
            self.nonlin = nonlin(**nonlin_kwargs)
            seq.append(self.nonlin)
        if nonlin_first and norm_op is not None and nonlin is not None:
            seq[-1], seq[-2] = seq[-2], seq[-1]
        self.all_modules = nn.Sequential(*seq)

    def forward(self, x):
        return self.all_modules(x)


class StackedConvBlocks(nn.Module):
    def __init__(self, num_convs, conv_op, input_channels, output_channels, kernel_size, initial_stride,
                 conv_bias=False, norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, nonlin_first=False):
        super().__init__()
        if not isinstance(output_channels, (tuple, list)):
            output_channels = [output_channels] * num_convs
        blk = lambda ci, co, st: ConvDropoutNormReLU(conv_op, ci, co, kernel_size, st, conv_bias, norm_op,
                                                     norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin,
                                                     nonlin_kwargs, nonlin_first)
        self.convs = nn.Sequential(blk(input_channels, output_channels[0], initial_stride),
                                   *[blk(output_channels[i - 1], output_channels[i], 1) for i in range(1, num_convs)])
        self.output_channels = output_channels[-1]
        self.initial_stride = _l3(initial_stride)

    def forward(self, x):
        return self.convs(x)


class PlainConvEncoder(nn.Module):
    def __init__(self, input_channels, n_stages, features_per_stage, conv_op, kernel_sizes, strides, n_conv_per_stage,
                 conv_bias=False, norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, return_skips=False, nonlin_first=False, pool="conv"):
        super().__init__()
        ex = lambda v: [v] * n_stages if isinstance(v, int) else v
        kernel_sizes, features_per_stage, n_conv_per_stage, strides = map(ex, (kernel_sizes, features_per_stage,
                                                                               n_conv_per_stage, strides))
        stages = []
        for s in range(n_stages):
            stages.append(nn.Sequential(StackedConvBlocks(n_conv_per_stage[s], conv_op, input_channels,
                                                          features_per_stage[s], kernel_sizes[s], strides[s], conv_bias,
                                                          norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin,
                                                          nonlin_kwargs, nonlin_first)))
            input_channels = features_per_stage[s]
        self.stages = nn.Sequential(*stages)
        self.output_channels = features_per_stage
        self.strides = [_l3(s) for s in strides]
        self.return_skips = return_skips
        self.conv_op, self.norm_op, self.norm_op_kwargs = conv_op, norm_op, norm_op_kwargs
        self.nonlin, self.nonlin_kwargs = nonlin, nonlin_kwargs
        self.dropout_op, self.dropout_op_kwargs = dropout_op, dropout_op_kwargs
        self.conv_bias, self.kernel_sizes = conv_bias, kernel_sizes

    def forward(self, x):
        ret = []
        for s in self.stages:
            x = s(x)
            ret.append(x)
        return ret if self.return_skips else ret[-1]


class UNetDecoder(nn.Module):
    def __init__(self, encoder, num_classes, n_conv_per_stage, deep_supervision, nonlin_first=False):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.encoder = encoder
        self.num_classes = num_classes
        n_enc = len(encoder.output_channels)
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * (n_enc - 1)
        stages, transpconvs, seg_layers = [], [], []
        for s in range(1, n_enc):
            below, skip = encoder.output_channels[-s], encoder.output_channels[-(s + 1)]
            st = encoder.strides[-s]
            transpconvs.append(nn.ConvTranspose3d(below, skip, st, st, bias=encoder.conv_bias))
            stages.append(StackedConvBlocks(n_conv_per_stage[s - 1], encoder.conv_op, 2 * skip, skip,
                                            encoder.kernel_sizes[-(s + 1)], 1, encoder.conv_bias, encoder.norm_op,
                                            encoder.norm_op_kwargs, encoder.dropout_op, encoder.dropout_op_kwargs,
                                            encoder.nonlin, encoder.nonlin_kwargs, nonlin_first))
            seg_layers.append(encoder.conv_op(skip, num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(transpconvs)
        self.seg_layers = nn.ModuleList(seg_layers)


class PlainConvUNet(nn.Module):
    def __init__(self, input_channels, n_stages, features_per_stage, conv_op, kernel_sizes, strides, n_conv_per_stage,
                 num_classes, n_conv_per_stage_decoder, conv_bias=False, norm_op=None, norm_op_kwargs=None,
                 dropout_op=None, dropout_op_kwargs=None, nonlin=None, nonlin_kwargs=None, deep_supervision=False,
                 nonlin_first=False):
        super().__init__()
        self.encoder = PlainConvEncoder(input_channels, n_stages, features_per_stage, conv_op, kernel_sizes, strides,
                                        n_conv_per_stage, conv_bias, norm_op, norm_op_kwargs, dropout_op,
                                        dropout_op_kwargs, nonlin, nonlin_kwargs, return_skips=True,
                                        nonlin_first=nonlin_first)
        self.decoder = UNetDecoder(self.encoder, num_classes, n_conv_per_stage_decoder, deep_supervision,
                                   nonlin_first=nonlin_first)

    def forward(self, x):
        return self.decoder(self.encoder(x))


class MemoryEfficientSoftDiceLoss(nn.Module):
    """-mean_{b,c} (2*sum(p*onehot) + smooth) / clip(sum(onehot) + sum(p) + smooth, 1e-8); batch_dice / ddp unused
    by the reference (utils/seg_utils.py:356-357)."""

    def __init__(self, apply_nonlin=None, batch_dice=False, do_bg=True, smooth=1.0, ddp=True):
        super().__init__()
        assert not batch_dice
        self.apply_nonlin, self.do_bg, self.smooth = apply_nonlin, do_bg, smooth

    def forward(self, x, y, loss_mask=None):
        assert loss_mask is None
        if self.apply_nonlin is not None:
            x = self.apply_nonlin(x)
        axes = list(range(2, x.ndim))
        with torch.no_grad():
            if x.ndim != y.ndim:
                y = y.view((y.shape[0], 1, *y.shape[1:]))
            onehot = torch.zeros(x.shape, dtype=torch.bool)
            onehot.scatter_(1, y.long(), 1)
            if not self.do_bg:
                onehot = onehot[:, 1:]
            sum_gt = onehot.sum(axes)
        if not self.do_bg:
            x = x[:, 1:]
        inter, sum_pred = (x * onehot).sum(axes), x.sum(axes)
        return -((2 * inter + self.smooth) / torch.clip(sum_gt + sum_pred + self.smooth, 1e-8)).mean()


def register_stand_ins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    mod("dynamic_network_architectures.architectures.unet", PlainConvUNet=PlainConvUNet)
    mod("dynamic_network_architectures.building_blocks.unet_decoder", UNetDecoder=UNetDecoder)
    mod("nnunetv2.training.loss.dice", MemoryEfficientSoftDiceLoss=MemoryEfficientSoftDiceLoss,
        SoftDiceLoss=MemoryEfficientSoftDiceLoss)
    mod("nnunetv2.utilities.helpers", softmax_helper_dim1=lambda x: torch.softmax(x, 1))


def import_reference_segmodel():
    register_stand_ins()
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference")
    import models.seg_model as sm
    import utils.seg_utils as su
    return sm, su


# ----------------------------------------------------------------------------- cases
SMALL = dict(n_stages=3, features_per_stage=[32, 64, 96], kernel_sizes=[[1, 3, 3], [3, 3, 3], [3, 3, 3]],
             strides=[[1, 1, 1], [1, 2, 2], [2, 2, 2]], n_conv_per_stage=[2, 2, 2], n_conv_per_stage_decoder=[2, 2],
             num_classes=2, upscale=4)
# distillation-compatible stride pattern (stage-1 stride (1,2,2)) in four stages
ANISO4 = dict(n_stages=4, features_per_stage=[32, 64, 128, 160],
              kernel_sizes=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3]],
              strides=[[1, 1, 1], [1, 2, 2], [1, 2, 2], [2, 2, 2]], n_conv_per_stage=[2, 2, 2, 2],
              n_conv_per_stage_decoder=[2, 2, 2], num_classes=2, upscale=4)
FULL_GRADS = ("sr_head.2.bias", "sr_head.0.bias", "decoder.seg_layers.1.weight", "decoder.transpconvs.0.bias",
              "encoder.stages.0.0.convs.0.conv.weight", "encoder.stages.1.0.convs.1.norm.weight",
              "decoder.stages.1.convs.0.norm.bias", "decoder.seg_layers.2.weight")


def canonical(key):
    if key.startswith("decoder.encoder."):
        key = key[len("decoder."):]
    return key.replace("all_modules.0.", "conv.").replace("all_modules.1.", "norm.")


def build_ref(sm, cfg, deep_supervision):
    m = sm.SegModel(input_channels=1, num_classes=cfg["num_classes"], n_stages=cfg["n_stages"],
                    upscale=cfg["upscale"], features_per_stage=cfg["features_per_stage"], conv_op=nn.Conv3d,
                    kernel_sizes=cfg["kernel_sizes"], strides=cfg["strides"],
                    n_conv_per_stage=cfg["n_conv_per_stage"],
                    n_conv_per_stage_decoder=cfg["n_conv_per_stage_decoder"], conv_bias=True,
                    norm_op=nn.InstanceNorm3d, norm_op_kwargs={"eps": 1e-5, "affine": True}, dropout_op=None,
                    dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs={"inplace": True},
                    deep_supervision=deep_supervision)
    m.load_state_dict({k: det_tensor(canonical(k), tuple(v.shape)) for k, v in m.state_dict().items()})
    return m


def segmodel_case(sm, su, tag, cfg, shape):
    """Stage-2 style step on the reference's SegModel (train_all.py:534,538-548): out / out_up / skips, the
    reference's DC + uncertainty-CE on the LR head and DC + CE on the HR head, gradients of every parameter."""
    m = build_ref(sm, cfg, False)
    N, _, D, H, W = shape
    x = det_input(tag + ".x", shape, "randn")
    lab_lr = det_input(tag + ".lab_lr", (N, 1, D, H, W), "randint2")
    lab_hr = det_input(tag + ".lab_hr", (N, 1, D * cfg["upscale"], H, W), "randint2")
    unc = 1.0 - torch.floor(det_input(tag + ".unc", (N, 1, D, H, W), "rand") * 256) / 255.0 * 0.99
    out, out_up, skips = m(x.clone(), return_inetermediate_feature=True)
    loss_fn = su._build_loss(False, weight_dice=1)
    loss_lr = loss_fn(out, lab_lr, unc)
    loss_hr = loss_fn(out_up, lab_hr, None)
    g3 = det_input(tag + ".g3", tuple(skips[1].shape), "randn")
    loss = loss_lr + loss_hr + (skips[1] * g3).mean()
    loss.backward()
    rec = {"x": x.numpy(), "lab_lr": lab_lr.numpy(), "lab_hr": lab_hr.numpy(), "unc": unc.numpy(),
           "out": out.detach().numpy(), "out_up": out_up.detach().numpy(),
           "loss": np.float64(loss.item()), "loss_lr": np.float64(loss_lr.item()), "loss_hr": np.float64(loss_hr.item())}
    for i, s in enumerate(skips):
        rec[f"skip{i}_shape"] = np.array(s.shape)
        rec[f"skip{i}_mean"] = s.detach().double().mean((2, 3, 4)).numpy()
    rec["skip1"] = skips[1].detach().numpy()
    names, norms = [], []
    seen = set()
    for k, p in m.named_parameters():      # named_parameters() lists every shared tensor once
        if p.grad is None or canonical(k) in seen:
            continue
        seen.add(canonical(k))
        names.append(canonical(k))
        norms.append(float(p.grad.double().norm()))
        if canonical(k) in FULL_GRADS:
            rec["grad:" + canonical(k)] = p.grad.numpy()
    rec["grad_names"], rec["grad_norms"] = np.array(names), np.array(norms, dtype=np.float64)
    # deep supervision on: list of LR logits, finest first (MyUnetDecoder.forward :40-51), same weights
    mds = build_ref(sm, cfg, True)
    with torch.no_grad():
        outs, up2 = mds(x.clone())
    assert isinstance(outs, list)
    for i, o in enumerate(outs):
        rec[f"ds_out{i}"] = o.numpy()
    rec["ds_out_up_maxdiff"] = np.float64((up2 - out_up.detach()).abs().max().item())
    np.savez_compressed(os.path.join(OUT, f"segmodel_{tag}.npz"), **rec)
    print(tag, "loss", loss.item(), "lr", loss_lr.item(), "hr", loss_hr.item(), "grads", len(names),
          "ds outs", [tuple(o.shape) for o in outs])


def loss_cases(su):
    """DC_and_weighted_CE_loss.forward (utils/seg_utils.py:306-351) as _build_loss builds it: values and logit
    gradients with / without the uncertainty map, two and three classes, weight_dice 1 and 0.5."""
    rec = {}
    for C in (2, 3):
        for with_unc in (False, True):
            for wd in (1.0, 0.5):
                tag = f"c{C}_u{int(with_unc)}_w{int(wd * 10)}"
                N, D, H, W = 2, 6, 10, 12
                lg = (det_input("loss.x." + tag, (N, C, D, H, W)) * 2.0).requires_grad_()
                tg = torch.floor(det_input("loss.t." + tag, (N, 1, D, H, W), "rand") * C).clamp_(0, C - 1)
                un = (1.0 - torch.floor(det_input("loss.u." + tag, (N, 1, D, H, W), "rand") * 256) / 255.0 * 0.99) \
                    if with_unc else None
                val = su._build_loss(False, weight_dice=wd)(lg, tg, un)
                val.backward()
                rec[tag + ".logits"] = lg.detach().numpy()
                rec[tag + ".target"] = tg.numpy()
                if with_unc:
                    rec[tag + ".unc"] = un.numpy()
                rec[tag + ".loss"] = np.float64(val.item())
                rec[tag + ".grad"] = lg.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "seg_losses.npz"), **rec)
    print("seg_losses:", {k: float(v) for k, v in rec.items() if k.endswith(".loss")})


def main():
    os.makedirs(OUT, exist_ok=True)
    sm, su = import_reference_segmodel()
    segmodel_case(sm, su, "small", SMALL, (2, 1, 4, 16, 16))
    segmodel_case(sm, su, "aniso4", ANISO4, (1, 1, 8, 32, 32))
    loss_cases(su)


if __name__ == "__main__":
    main()
