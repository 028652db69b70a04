"""CPU restatements of the rows next to the conv path (TEST INFRASTRUCTURE): the
distillation loss, the in-reference losses, zscore_normalization and the teacher pass.
Pinned by tests/golden/aux_losses_teacher.npz (tools/gen_golden_losses.py ran the reference).

  distiller_loss      models/seg_model.py:60-151
  bce_dice            utils/seg_utils.py:786-885
  robust_ce           utils/seg_utils.py:289-304 (incl. the (B,B,...) uncertainty broadcast, :349)
  dc_and_weighted_ce  utils/seg_utils.py:306-351 as _build_loss (:353-357) configures it; the composition is pinned by
                      tests/golden/seg_losses.npz (tools/gen_golden_segmodel.py ran the reference's forward over a
                      stand-in for nnunetv2==2.3.1's MemoryEfficientSoftDiceLoss, which is absent offline: the
                      soft-Dice formula itself stays PARITY UNPINNED)
  zscore              utils/seg_utils.py:137-149 (in place)
  teacher_features    train_all.py:85-112 (one window at a time, like the reference)
"""
import torch
import torch.nn.functional as F

from . import flavr_oracle as fo


def distiller_loss(w, b, fs, ft, lambda_l1, lambda_cosine, lambda_structure, emu=None):
    """emu (oracle/bf16_emul.py): the 1x1x1 projection takes bf16 operands and stores bf16 on the mixed-precision path;
    the structure / cosine statistics are fp32 there (on the bf16 feature values)."""
    loss = 0
    if emu is not None:
        fs = emu.act(fs)
    if lambda_structure > 0:
        Bn, C, S, Hh, Ww = fs.shape
        to2d = lambda t: t.permute(0, 2, 1, 3, 4).reshape(Bn * S, C, Hh, Ww)
        kh, kw = int(Hh * 0.5), int(Ww * 0.5)
        pool = lambda t: F.max_pool2d(to2d(t), (kh, kw), (kh, kw), 0, ceil_mode=True)

        def gram(f):
            n = f.pow(2).sum(1, keepdim=True).sqrt() + 1e-8
            f = (f / n.detach()).flatten(2)
            return torch.einsum("icm,icn->imn", f, f)
        ps, pt = pool(fs), pool(ft)
        err = (gram(pt) - gram(ps)).pow(2) / ((pt.shape[-1] * pt.shape[-2]) ** 2) / pt.shape[0]
        loss = loss + lambda_structure * err.sum() / S
    d = F.conv3d(fs, w, b) if emu is None else emu.act(F.conv3d(fs, emu.weight(w), b))
    if lambda_l1 > 0:
        loss = loss + lambda_l1 * F.smooth_l1_loss(d, ft)
    if lambda_cosine > 0:
        a = F.normalize(d, p=2, dim=1).flatten(2)
        c = F.normalize(ft, p=2, dim=1).flatten(2)
        loss = loss + lambda_cosine * (1 - F.cosine_similarity(a, c, dim=2)).mean()
    return loss


def bce_dice(logits, target, alpha=1.0, beta=1.0):
    p = torch.sigmoid(logits)
    c = p.shape[1]
    pf, tf = p.transpose(0, 1).reshape(c, -1), target.transpose(0, 1).reshape(c, -1).float()
    dice = 2 * (pf * tf).sum(-1) / ((pf * pf).sum(-1) + (tf * tf).sum(-1)).clamp(min=1e-6)
    return alpha * F.binary_cross_entropy_with_logits(logits, target) + beta * (1.0 - dice.mean())


def robust_ce(logits, target, uncertainty=None):
    loss = F.cross_entropy(logits, target.long(), reduction="none")   # (B, D, H, W)
    if uncertainty is not None:
        loss = loss * uncertainty                                        # (B,1,D,H,W) -> broadcast (B,B,D,H,W)
    return loss.mean()


def soft_dice(logits, target, smooth=1e-5, do_bg=False):
    """nnunetv2 MemoryEfficientSoftDiceLoss (batch_dice=False), restated -- unpinned third-party formula."""
    p = torch.softmax(logits, 1)
    onehot = F.one_hot(target[:, 0].long(), logits.shape[1]).movedim(-1, 1).to(p.dtype)
    if not do_bg:
        p, onehot = p[:, 1:], onehot[:, 1:]
    axes = tuple(range(2, p.ndim))
    dc = (2 * (p * onehot).sum(axes) + smooth) / torch.clip(onehot.sum(axes) + p.sum(axes) + smooth, 1e-8)
    return -dc.mean()


def dc_and_weighted_ce(logits, target, uncertainty=None, weight_ce=1.0, weight_dice=1.0):
    """target (B,1,D,H,W) float labels; uncertainty (B,1,D,H,W) or None."""
    return weight_ce * robust_ce(logits, target[:, 0], uncertainty) + weight_dice * soft_dice(logits, target)


def zscore(image):
    outs = []
    for i in range(image.shape[0]):
        v = image[i:i + 1, 0]
        m, s = v.mean(), v.std()
        v -= m
        v /= max(s, 1e-8)
        outs.append(v)
    return torch.stack(outs, 0)


def teacher_features(sd, img_lr, label_lr, img_channels=2, n_inputs=4, n_outputs=4, use_uncertainty=True, emu=None,
                     upto=4):
    """upto < 4 stops every window's encoder after that level (the levels up to it are unchanged: the encoder is a
    chain) -- the full-size tests only need level 1 and the CPU time of 127 windows matters there."""
    img_lr = zscore(img_lr)
    x = torch.cat((img_lr, label_lr), 1)
    D = x.shape[2]
    per = {}
    last = None
    for st in range(D - 1):
        if st == 0:
            w = torch.cat([torch.zeros_like(x[:, :, :1]), x[:, :, 0:3]], 2)
        elif st == D - 2:
            w = x[:, :, st - 1:]
            w = torch.cat([w, torch.zeros(w.shape[0], w.shape[1], 4 - w.shape[2], *w.shape[3:])], 2)
        else:
            w = x[:, :, st - 1:st + 3]
        last = fo.unet_3d_3d(sd, w.clone(), img_channels, n_inputs, n_outputs, use_uncertainty,
                             return_intermediate_feature=True, emu=emu, upto=upto)
        for i, f in enumerate(last):
            per.setdefault(i, []).append(f[:, :, 1:2])
    for i, f in enumerate(last):
        per[i].append(f[:, :, 2:3])
    return {i: torch.cat(v, 2) for i, v in per.items()}
