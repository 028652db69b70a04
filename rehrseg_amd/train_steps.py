"""The two hot training loops of the reference as per-step functions, plus the teacher
pass, on the MI355X modules.

  get_intermediate_features   train_all.py:85-112   (teacher features for distillation)
  train_sr_step               train_all.py:118-139  (stage 1b/1c: FLAVR self-SR step)
  train_segsr_step            train_all.py:521-556  (stage 2: SegModel + distillation step)

Results are those of the reference loops; two things are restructured for the GPU:
  * the teacher's D-1 four-slice windows are ONE batched encoder call instead of D-1
    calls (windows are independent samples: no BatchNorm, per-sample SE pooling), and
  * `levels` lets the caller stop the teacher after the level it consumes (the stage-2
    loop only reads level 1).
"""
import torch
import torch.nn.functional as F

from .utils.seg_utils import zscore_normalization


def get_intermediate_features(model_sr, img_lr, label_lr, device=None, levels=None, _normalized=None):
    """dict level -> (B, C_level, D, h, w).  Like the reference it z-scores `img_lr` IN PLACE
    (the student is fed the normalised image afterwards, train_all.py:533-534).
    _normalized: the result of zscore_normalization(img_lr) when the caller has already applied it (train_segsr_step
    normalises on the main stream and runs the rest of the teacher pass on a second one)."""
    img_lr = zscore_normalization(img_lr) if _normalized is None else _normalized
    x = torch.cat((img_lr, label_lr), dim=1)                     # (B, 2, D, H, W)
    B, C, D, H, W = x.shape
    if D < 2:
        raise ValueError("need at least two slices")
    padded = F.pad(x, (0, 0, 0, 0, 1, 2))                       # [0, x_0 .. x_{D-1}, 0, 0] along depth
    upto = 4 if levels is None else max(levels)
    enc = getattr(model_sr, "encoder", None)
    truncated = upto < 4 and enc is not None and "upto" in enc.forward.__code__.co_varnames
    if truncated and not torch.is_grad_enabled() and H % 2 == 0 and W % 2 == 0:
        # frozen teacher: the stem's per-slice work is shared between the overlapping windows
        from .models.FLAVR.resnet_3D import encoder_on_windows
        feats = encoder_on_windows(enc, padded[:, :, :D + 2], D - 1, upto)  # (the windows never reach the last slice)
    else:
        # window st = padded[st : st+4], st = 0 .. D-2 (zero slice in front of the first, behind the last)
        win = padded.unfold(2, 4, 1)[:, :, :D - 1]              # (B, 2, D-1, H, W, 4)
        win = win.permute(0, 2, 1, 5, 3, 4).reshape(B * (D - 1), C, 4, H, W).contiguous()
        if truncated:
            mean_ = win[:, 0:1].mean(2, keepdim=True).mean(3, keepdim=True).mean(4, keepdim=True)
            win[:, 0:1] = win[:, 0:1] - mean_                   # UNet_3D_3D.forward's mean subtraction
            feats = enc(win, upto=upto)
        else:
            feats = model_sr(win, return_inetermediate_feature=True)
    out = {}
    for i, f in enumerate(feats):
        if levels is not None and i not in levels:
            continue
        f = f.reshape(B, D - 1, f.shape[1], 4, f.shape[3], f.shape[4])
        mid = f[:, :, :, 1].permute(0, 2, 1, 3, 4)              # slice 1 of every window
        last = f[:, -1, :, 2].unsqueeze(2)                      # slice 2 of the last window
        out[i] = torch.cat([mid, last], dim=2)
    return out


def _zero_grad(opt, grad_sync, zero_grad):
    """`opt.zero_grad()` of the reference loops.  With patch-parallel training (grad_sync =
    PatchParallel.reduce_gradients) the wrapper's own zero_grad is used: it keeps the conv weights' `.grad`
    views into the flat exchange buffer, so their gradients are written in place instead of re-allocated.
    (A plain optimizer.zero_grad() stays correct -- PatchParallel reconciles at the exchange -- just slower.)"""
    if zero_grad is not None:
        return zero_grad()
    owner = getattr(grad_sync, "__self__", None)
    if owner is not None and callable(getattr(owner, "zero_grad", None)) and hasattr(owner, "reduce_gradients"):
        return owner.zero_grad()
    return opt.zero_grad()


def train_sr_step(model, opt, scheduler, patches_lr, patches_hr, loss_obj, loss_seg, slice_separation, num_slices,
                  enable_uncertainty, grad_sync=None, zero_grad=None):
    """One iteration of train_sr's inner loop (train_all.py:118-139); returns the loss tensor.
    grad_sync: called between backward and the optimizer step (PatchParallel.reduce_gradients);
    zero_grad: overrides how gradients are cleared (default: see _zero_grad)."""
    if num_slices > 1:
        s = int(slice_separation)
        patches_hr = patches_hr[:, :, s * (num_slices // 2 - 1):s * (num_slices // 2), ...]
    if enable_uncertainty:
        hat, unc = model(patches_lr)
        loss = loss_obj(hat[:, 0:1], patches_hr[:, 0:1])
        loss = loss + torch.mean(torch.div(torch.abs(hat[:, 0:1] - patches_hr[:, 0:1]), unc) + torch.log(unc))
        loss = loss + loss_obj(unc, torch.abs(hat[:, 0:1].detach() - patches_hr[:, 0:1]))
    else:
        hat = model(patches_lr)
        loss = loss_obj(hat[:, 0:1], patches_hr[:, 0:1])
    loss = loss + loss_seg(hat[:, 1:], patches_hr[:, 1:]) * 1.0
    _zero_grad(opt, grad_sync, zero_grad)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    opt.step()
    if scheduler is not None:
        scheduler.step()
    return loss


_TEACHER_STREAMS = {}


def _teacher_stream(t):
    """One extra HIP stream per device for the teacher pass of train_segsr_step (None for CPU tensors: host-logic tests)."""
    if not t.is_cuda:
        return None
    key = t.device.index
    if key not in _TEACHER_STREAMS:
        _TEACHER_STREAMS[key] = torch.cuda.Stream(device=t.device)
    return _TEACHER_STREAMS[key]


def train_segsr_step(model_seg, model_sr, distiller, opt, img, label_lr, label_hr, uncertainty_lr, loss_lr_seg,
                     loss_hr_seg, enable_uncertainty=True, teacher_levels=(1,), grad_sync=None, zero_grad=None,
                     teacher_stream=True):
    """One iteration of the stage-2 loop (train_all.py:521-556); returns the loss tensor.
    teacher_stream: run the frozen teacher's pass on a second HIP stream next to the student's forward."""
    model_seg.train()
    if distiller is not None:
        side = _teacher_stream(img) if teacher_stream else None
        if side is None:
            with torch.no_grad():
                features_sr = get_intermediate_features(model_sr, img, label_lr, img.device, levels=teacher_levels)
            seg_lr, seg_sr, features_seg = model_seg(img, return_inetermediate_feature=True)
        else:
            # The frozen teacher's pass and the student's forward are independent once the image is z-scored (in place,
            # as the reference does: the student reads the normalised image): the teacher runs on a second HIP stream,
            # the student on the current one, joined before the distillation loss.  Same kernels, same results.
            main = torch.cuda.current_stream(img.device)
            with torch.no_grad():
                normalized = zscore_normalization(img)
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                features_sr = get_intermediate_features(model_sr, img, label_lr, img.device, levels=teacher_levels,
                                                        _normalized=normalized)
            seg_lr, seg_sr, features_seg = model_seg(img, return_inetermediate_feature=True)
            main.wait_stream(side)
            for t in features_sr.values():
                t.record_stream(main)          # allocated on the side stream, consumed (and later freed) on the main one
            for t in (img, label_lr, normalized):
                t.record_stream(side)
    else:
        seg_lr, seg_sr = model_seg(img)
    if enable_uncertainty:
        loss = loss_lr_seg(seg_lr, label_lr, uncertainty_lr) + loss_hr_seg(seg_sr, label_hr, None)
    else:
        loss = loss_lr_seg(seg_lr, label_lr) + loss_hr_seg(seg_sr, label_hr)
    if distiller is not None:
        loss = loss + distiller(features_seg[1], features_sr[1])
    _zero_grad(opt, grad_sync, zero_grad)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    opt.step()
    return loss
