"""Losses and normalisation used by the two training loops, under the reference's import
path (`utils.seg_utils`): same names, arguments and numerics as utils/seg_utils.py:137-156
(zscore_normalization), :289-372 (RobustCrossEntropyLoss, DC_and_weighted_CE_loss,
_build_loss) and :786-885 (DiceLoss, BCEDiceLoss).

The stage-2 loss (`DC_and_weighted_CE_loss`) runs as one fused HIP pass over the logits each way when
given device tensors (`_FusedDCCE`); the torch composition below it is the same
arithmetic and is what the CPU-side tests and the oracle comparisons evaluate.  The tiled predictor
helpers at the end of the file mirror utils/seg_utils.py:176-287.
`MemoryEfficientSoftDiceLoss` lives in nnunetv2==2.3.1 (absent offline): its published
formula is restated in `SoftDiceLoss` below -- that term's parity is unpinned.
"""
import functools

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def zscore_normalization(image):
    """Per-sample z-score of channel 0, IN PLACE on the caller's tensor (ref :137-149);
    returns the normalised channel as (B, 1, ...)."""
    if not isinstance(image, torch.Tensor):
        image = image.astype(np.float32, copy=False)
        image -= image.mean()
        image /= max(image.std(), 1e-8)
        return image
    outs = []
    for i in range(image.shape[0]):
        view = image[i:i + 1, 0, ...]
        mean, std = view.mean(), view.std()
        view -= mean
        view /= max(std, 1e-8)
        outs.append(view)
    return torch.stack(outs, dim=0)


class RobustCrossEntropyLoss(nn.CrossEntropyLoss):
    """CE on float targets with an optional uncertainty weight (ref :289-304).  The weight has a
    channel axis the loss map lacks, so (B,D,H,W) * (B,1,D,H,W) broadcasts to (B,B,D,H,W)
    before the mean -- reproduced as is."""

    def forward(self, input, target, uncertainty=None):
        if target.ndim == input.ndim:
            assert target.shape[1] == 1
            target = target[:, 0]
        loss = super().forward(input, target.long())
        if uncertainty is not None:
            loss = loss * uncertainty
        return loss.mean()


class SoftDiceLoss(nn.Module):
    """nnunetv2 MemoryEfficientSoftDiceLoss, restated (unpinned): -mean over (b, c) of
    (2*intersect + smooth) / clip(sum_gt + sum_pred + smooth, 1e-8)."""

    def __init__(self, apply_nonlin=None, batch_dice=False, do_bg=True, smooth=1.0, ddp=False):
        super().__init__()
        if ddp and batch_dice:
            raise NotImplementedError("REHRSeg passes batch_dice=False, ddp=False (utils/seg_utils.py:356-357)")
        self.apply_nonlin, self.batch_dice, self.do_bg, self.smooth = apply_nonlin, batch_dice, do_bg, smooth

    def forward(self, x, y, loss_mask=None):
        if self.apply_nonlin is not None:
            x = self.apply_nonlin(x)
        axes = tuple(range(2, x.ndim))
        with torch.no_grad():
            if x.ndim != y.ndim:
                y = y.view((y.shape[0], 1, *y.shape[1:]))
            if x.shape == y.shape:
                y_onehot = y
            else:
                y_onehot = torch.zeros(x.shape, device=x.device, dtype=torch.bool)
                y_onehot.scatter_(1, y.long(), 1)
            if not self.do_bg:
                y_onehot = y_onehot[:, 1:]
            sum_gt = y_onehot.sum(axes) if loss_mask is None else (y_onehot * loss_mask).sum(axes)
        if not self.do_bg:
            x = x[:, 1:]
        if loss_mask is None:
            intersect, sum_pred = (x * y_onehot).sum(axes), x.sum(axes)
        else:
            intersect, sum_pred = (x * y_onehot * loss_mask).sum(axes), (x * loss_mask).sum(axes)
        if self.batch_dice:
            intersect, sum_pred, sum_gt = intersect.sum(0), sum_pred.sum(0), sum_gt.sum(0)
        dc = (2 * intersect + self.smooth) / torch.clip(sum_gt + sum_pred + self.smooth, 1e-8)
        return -dc.mean()


def softmax_helper_dim1(x):
    return torch.softmax(x, 1)


class _FusedDCCE(torch.autograd.Function):
    """softmax + (uncertainty-weighted) CE + soft Dice as one HIP pass over the logits each way
    (include/rehrseg_hip.h: rehr_seg_loss_{fwd,bwd}_f32)."""

    @staticmethod
    def forward(ctx, logits, target, unc, w_ce, w_dice, smooth, do_bg):
        from .. import hip_backend as hb, ops
        logits = ops.to_cl(logits)
        N, C = logits.shape[:2]
        S = logits.shape[2] * logits.shape[3] * logits.shape[4]
        target = target.reshape(N, S).to(torch.float32).contiguous()
        unc = None if unc is None else unc.reshape(N, S).to(torch.float32).contiguous()
        stats = hb.seg_loss_fwd(logits, target, unc)
        ipg = stats[:-1].view(N, C, 3)
        dc = (2 * ipg[..., 0] + smooth) / torch.clip(ipg[..., 2] + ipg[..., 1] + smooth, 1e-8)
        if not do_bg:
            dc = dc[:, 1:]
        loss = w_ce * stats[-1] / (N * N * S if unc is not None else N * S) - w_dice * dc.mean()
        ctx.save_for_backward(logits, target, unc, stats)
        ctx.cfg = (w_ce, w_dice, smooth, do_bg)
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        from .. import hip_backend as hb
        logits, target, unc, stats = ctx.saved_tensors
        w_ce, w_dice, smooth, do_bg = ctx.cfg
        go = grad_out.to(torch.float32).reshape(1).contiguous()
        return hb.seg_loss_bwd(logits, target, unc, stats, w_ce, w_dice, smooth, do_bg, go), None, None, None, \
            None, None, None


class DC_and_weighted_CE_loss(nn.Module):
    """ref :306-351: weight_ce * CE(uncertainty-weighted) + weight_dice * soft Dice."""
    _warned = False

    def __init__(self, soft_dice_kwargs, ce_kwargs, weight_ce=1, weight_dice=1, ignore_label=None,
                 dice_class=SoftDiceLoss):
        super().__init__()
        if ignore_label is not None:
            ce_kwargs["ignore_index"] = ignore_label
        self.weight_dice, self.weight_ce, self.ignore_label = weight_dice, weight_ce, ignore_label
        self.ce = RobustCrossEntropyLoss(**ce_kwargs)
        self.dc = dice_class(apply_nonlin=softmax_helper_dim1, **soft_dice_kwargs)

    def _fusable(self, net_output, target, uncertainty):
        """The fused kernel covers what the reference's stage-2 loop passes (train_all.py:538-548): 5-D logits with
        2..4 classes, a (N,1,D,H,W) label map with values in [0, C) (precondition, unchecked: torch's CE would raise),
        no ignore label, and an uncertainty map of exactly (N,1,D,H,W) -- the shape whose product with the (N,D,H,W)
        loss map broadcasts across samples; any other uncertainty shape multiplies differently
        in the reference and takes the composition below."""
        dc = self.dc
        N = net_output.shape[0]
        return (net_output.is_cuda and net_output.dim() == 5 and self.ignore_label is None and
                type(dc) is SoftDiceLoss and not dc.batch_dice and dc.apply_nonlin is softmax_helper_dim1 and
                2 <= net_output.shape[1] <= 4 and target.dim() == 5 and target.shape[1] == 1 and
                target.numel() == net_output.numel() // net_output.shape[1] and
                (uncertainty is None or tuple(uncertainty.shape) == (N, 1) + tuple(net_output.shape[2:])) and
                self.ce.weight is None and self.ce.label_smoothing == 0.0)

    def forward(self, net_output, target, uncertainty=None):
        if self._fusable(net_output, target, uncertainty):  # device tensors: one HIP pass each way
            return _FusedDCCE.apply(net_output, target, uncertainty, float(self.weight_ce), float(self.weight_dice),
                                    float(self.dc.smooth), bool(self.dc.do_bg))
        if net_output.is_cuda and not DC_and_weighted_CE_loss._warned:
            DC_and_weighted_CE_loss._warned = True
            import warnings
            warnings.warn("DC_and_weighted_CE_loss: configuration outside the fused HIP kernel (classes > 4, ignore label, "
                          "class weights or an unusual target / uncertainty shape): composed from torch ops on the device")
        if self.ignore_label is not None:
            assert target.shape[1] == 1
            mask = target != self.ignore_label
            target_dice = torch.where(mask, target, 0)
            num_fg = mask.sum()
        else:
            target_dice, mask = target, None
        dc_loss = self.dc(net_output, target_dice, loss_mask=mask) if self.weight_dice != 0 else 0
        ce_loss = self.ce(net_output, target[:, 0], uncertainty) \
            if self.weight_ce != 0 and (self.ignore_label is None or num_fg > 0) else 0
        return self.weight_ce * ce_loss + self.weight_dice * dc_loss


def _build_loss(enable_deep_supervision=False, weight_dice=1):
    """ref :353-372 without the DeepSupervisionWrapper branch (the reference trains with
    enable_deep_supervision=False, train_all.py:471)."""
    if enable_deep_supervision:
        raise NotImplementedError("deep supervision wrapper (nnunetv2) is not part of the hot path")
    return DC_and_weighted_CE_loss({"batch_dice": False, "smooth": 1e-5, "do_bg": False, "ddp": False},
                                   {"reduction": "none"}, weight_ce=1, weight_dice=weight_dice, ignore_label=None,
                                   dice_class=SoftDiceLoss)


def compute_per_channel_dice(input, target, epsilon=1e-6, weight=None):
    """ref :829-857: 2*sum(p*t) / clamp(sum(p^2) + sum(t^2), eps) per channel."""
    assert input.size() == target.size()
    c = input.size(1)
    p = input.transpose(0, 1).reshape(c, -1)
    t = target.transpose(0, 1).reshape(c, -1).float()
    inter = (p * t).sum(-1)
    if weight is not None:
        inter = weight * inter
    return 2 * (inter / ((p * p).sum(-1) + (t * t).sum(-1)).clamp(min=epsilon))


class DiceLoss(nn.Module):
    def __init__(self, weight=None, normalization="sigmoid"):
        super().__init__()
        self.register_buffer("weight", weight)
        assert normalization in ("sigmoid", "softmax", "none")
        self.normalization = {"sigmoid": torch.sigmoid, "softmax": lambda x: torch.softmax(x, 1),
                              "none": lambda x: x}[normalization]

    def forward(self, input, target):
        return 1.0 - torch.mean(compute_per_channel_dice(self.normalization(input), target, weight=self.weight))


class _FusedBCEDice(torch.autograd.Function):
    """BCE-with-logits + sigmoid Dice as one HIP pass over the logits each way (rehr_bce_dice_{fwd,bwd}_f32)."""

    @staticmethod
    def forward(ctx, logits, target, alpha, beta):
        from .. import hip_backend as hb
        N, C = logits.shape[:2]
        x = logits.reshape(N, C, -1).to(torch.float32).contiguous()
        t = target.reshape(N, C, -1).to(torch.float32).contiguous()
        stats = hb.bce_dice_fwd(x, t)
        dice = 2 * stats[:, 1] / (stats[:, 2] + stats[:, 3]).clamp(min=1e-6)
        loss = alpha * stats[:, 0].sum() / x.numel() + beta * (1.0 - dice.mean())
        ctx.save_for_backward(x, t, stats)
        ctx.cfg = (alpha, beta, tuple(logits.shape))
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        from .. import hip_backend as hb
        x, t, stats = ctx.saved_tensors
        alpha, beta, shape = ctx.cfg
        go = grad_out.to(torch.float32).reshape(1).contiguous()
        return hb.bce_dice_bwd(x, t, stats, alpha, beta, go).reshape(shape), None, None, None


class BCEDiceLoss(nn.Module):
    """alpha * BCEWithLogits + beta * Dice (ref :872-885).  Device tensors take one fused HIP pass each way."""

    def __init__(self, alpha, beta):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        self.bce = nn.BCEWithLogitsLoss()
        self.dice = DiceLoss()

    def forward(self, input, target):
        if input.is_cuda and input.dim() >= 3 and input.shape == target.shape and self.dice.weight is None:
            return _FusedBCEDice.apply(input, target, float(self.alpha), float(self.beta))
        return self.alpha * self.bce(input, target) + self.beta * self.dice(input, target)


# ----------------------------------------------------------------------------- whole-volume inference
# (ref utils/seg_utils.py:176-287: nnU-Net style tiled predictor with mirror TTA)
def compute_steps_for_sliding_window(image_size, tile_size, tile_step_size):
    """ref :176-199."""
    assert [i >= j for i, j in zip(image_size, tile_size)], "image size must be as large or larger than patch_size"
    assert 0 < tile_step_size <= 1, "step_size must be larger than 0 and smaller or equal to 1"
    target = [i * tile_step_size for i in tile_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, tile_size)]
    steps = []
    for dim in range(len(tile_size)):
        max_step = image_size[dim] - tile_size[dim]
        actual = max_step / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def _internal_get_sliding_window_slicers(image_size, patch_size=[14, 320, 384], tile_step_size=0.5):
    """ref :229-238."""
    steps = compute_steps_for_sliding_window(image_size, patch_size, tile_step_size)
    return [tuple([slice(None), *[slice(si, si + ti) for si, ti in zip((sx, sy, sz), patch_size)]])
            for sx in steps[0] for sy in steps[1] for sz in steps[2]]


def _internal_maybe_mirror_and_predict(model, x, out_idx=None, deep_supervision=True, save=False):
    """ref :201-227: mean over the identity and the 7 mirrorings of (D,H,W).  The 8 variants are independent
    (InstanceNorm is per sample), so they go through the network as ONE batch instead of 8 calls; the
    un-mirrored predictions are summed in the reference's order."""
    import itertools
    mirror_axes = (0, 1, 2)
    assert max(mirror_axes) <= x.ndim - 3, "mirror_axes does not match the dimension of the input!"
    combos = [c for i in range(len(mirror_axes)) for c in itertools.combinations([m + 2 for m in mirror_axes], i + 1)]
    B = x.shape[0]
    pred = model(torch.cat([x] + [torch.flip(x, axes) for axes in combos], 0))
    if out_idx is not None:
        pred = pred[out_idx]
        if out_idx == 0 and deep_supervision:
            pred = pred[0]
    prediction = pred[:B].clone()
    for k, axes in enumerate(combos):
        prediction += torch.flip(pred[(k + 1) * B:(k + 2) * B], axes)
    prediction /= (len(combos) + 1)
    return prediction


@functools.lru_cache(maxsize=2)
def compute_gaussian(tile_size, sigma_scale=1. / 8, value_scaling_factor=1, dtype=torch.float16, device="cpu"):
    """nnunetv2 compute_gaussian, restated (absent offline -> unpinned): Gaussian importance map of a tile."""
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(tile_size)
    tmp[tuple(i // 2 for i in tile_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode="constant", cval=0)
    g = torch.from_numpy(g)
    g = g / (torch.max(g) / value_scaling_factor)
    g = g.to(device=device, dtype=dtype)
    mask = g == 0
    g[mask] = torch.min(g[~mask])
    return g


def _internal_predict_sliding_window_return_logits(data, slicers, network, do_on_device=True, out_idx=None,
                                                   slice_seperation=1, patch_size=[14, 320, 384],
                                                   use_gaussian=False, deep_supervision=True):
    """ref :240-287 (fp16 accumulators as there; (sic) `slice_seperation`)."""
    results_device = data.device if do_on_device else torch.device("cpu")
    if do_on_device and not data.is_cuda and torch.cuda.is_available():
        results_device = torch.device("cuda")
    data = data.to(results_device)
    predicted_logits = torch.zeros((2, data.shape[1] * slice_seperation, data.shape[2], data.shape[3]),
                                   dtype=torch.half, device=results_device)
    n_predictions = torch.zeros((data.shape[1] * slice_seperation, data.shape[2], data.shape[3]), dtype=torch.half,
                                device=results_device)
    gaussian = compute_gaussian(tuple(patch_size), sigma_scale=1. / 8, value_scaling_factor=10,
                                device=str(results_device)) if use_gaussian else 1
    for i, sl in enumerate(slicers):
        workon = data[sl][None].to(results_device)
        prediction = _internal_maybe_mirror_and_predict(network, workon, out_idx, deep_supervision, i == len(slicers) - 1)
        prediction = prediction[0].to(results_device)
        msl = tuple([slice(None)] + [slice(sl[k].start * slice_seperation, sl[k].stop * slice_seperation) if k == 1
                                     else slice(sl[k].start, sl[k].stop) for k in range(1, len(sl))])
        predicted_logits[msl] += prediction * gaussian
        n_predictions[msl[1:]] += gaussian
    predicted_logits /= n_predictions
    if torch.any(torch.isinf(predicted_logits)):
        raise RuntimeError("Encountered inf in predicted array. Aborting... If this problem persists, reduce "
                           "value_scaling_factor in compute_gaussian or increase the dtype of predicted_logits to fp32")
    if use_gaussian:
        compute_gaussian.cache_clear()
    return predicted_logits
