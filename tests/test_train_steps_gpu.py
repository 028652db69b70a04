"""The two training steps end to end on the MI355X modules (losses, teacher pass, distillation,
optimizer) against the same step composed from the CPU oracles."""
import pytest
import torch
import torch.nn as nn

from oracle import aux_oracle as ao
from oracle import flavr_oracle as fo
from oracle import segmodel_oracle as so
from oracle.detinit import det_input, det_tensor
from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
from rehrseg_amd.models.seg_model import Distiller
from rehrseg_amd.train_steps import train_segsr_step, train_sr_step
from rehrseg_amd.utils import seg_utils as su
from test_segmodel_cpu import build, canonical

pytestmark = pytest.mark.gpu
PLAN = dict(n_stages=3, features_per_stage=[32, 64, 96], kernel_sizes=[[1, 3, 3], [1, 3, 3], [3, 3, 3]],
            strides=[[1, 1, 1], [1, 2, 2], [2, 2, 2]], n_conv_per_stage=[2, 2, 2], n_conv_per_stage_decoder=[2, 2],
            num_classes=2, upscale=4)


def _flavr(unc, dev):
    m = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=unc)
    sd = {k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    return m.to(dev), sd


def test_train_sr_step_uncertainty():
    dev = torch.device("cuda:0")
    m, sd = _flavr(True, dev)
    lr_p = det_input("sr.lr", (2, 2, 4, 32, 32), "rand")
    hr_p = det_input("sr.hr", (2, 2, 16, 32, 32), "rand")
    hr_p[:, 1:] = (hr_p[:, 1:] > 0.5).float()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    w_before = m.outconv[1].weight.detach().clone()
    loss = train_sr_step(m, opt, None, lr_p.clone().to(dev), hr_p.to(dev), nn.L1Loss(), su.BCEDiceLoss(1.0, 1.0), 4.0, 4,
                         True)
    # oracle composition of train_all.py:122-134
    osd = {k: v.clone().requires_grad_() for k, v in sd.items()}
    hat, unc = fo.unet_3d_3d(osd, lr_p.clone(), 2, 4, 4, True)
    hr = hr_p[:, :, 4:8]
    ref = (hat[:, 0:1] - hr[:, 0:1]).abs().mean()
    ref = ref + torch.mean((hat[:, 0:1] - hr[:, 0:1]).abs() / unc + torch.log(unc))
    ref = ref + (unc - (hat[:, 0:1].detach() - hr[:, 0:1]).abs()).abs().mean()
    ref = ref + ao.bce_dice(hat[:, 1:], hr[:, 1:])
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item())
    # the gradients the step's backward left behind (train_all.py:137) against the oracle composition's
    ref.backward()
    worst = ("", 0.0)
    for k, prm in m.named_parameters():
        if prm.grad is None or osd[k].grad is None:
            continue
        n = float(osd[k].grad.norm())
        if n == 0.0:
            continue
        e = float((prm.grad.cpu() - osd[k].grad).norm()) / n
        worst = max(worst, (k, e), key=lambda t: t[1])
    print("train_sr step: worst gradient l2-rel vs the oracle", worst)
    assert worst[1] <= 1e-3, worst
    assert torch.equal(m.outconv[1].weight.detach().cpu(), w_before.cpu())  # unused by the UASR head: no update
    assert not torch.equal(m.feature_fuse1.conv[0].weight.detach().cpu(), sd["feature_fuse1.conv.0.weight"])


def test_train_segsr_step_with_distillation():
    dev = torch.device("cuda:0")
    teacher, tsd = _flavr(True, dev)
    teacher.eval()
    student, ssd = build(PLAN, dev)
    dist = Distiller(64, 64, 0.0, 1.0, 1.0)
    dsd = {k: det_tensor(k, tuple(v.shape)) for k, v in dist.state_dict().items()}
    dist.load_state_dict(dsd)
    dist = dist.to(dev)
    img = det_input("st2.img", (2, 1, 6, 32, 32), "rand") * 2 + 0.5
    lab_lr = det_input("st2.lr", (2, 1, 6, 32, 32), "randint2")
    lab_hr = det_input("st2.hr", (2, 1, 24, 32, 32), "randint2")
    unc = 1 - det_input("st2.u", (2, 1, 6, 32, 32), "rand") * 0.99
    import itertools
    opt = torch.optim.SGD(itertools.chain(student.parameters(), dist.parameters()), lr=1e-3, momentum=0.99,
                          nesterov=True, weight_decay=3e-5)
    img_dev = img.clone().to(dev)
    loss = train_segsr_step(student, teacher, dist, opt, img_dev, lab_lr.to(dev), lab_hr.to(dev), unc.to(dev),
                            su._build_loss(False, weight_dice=0), su._build_loss(False, weight_dice=1))
    # oracle composition of train_all.py:531-552
    img_o = img.clone()
    with torch.no_grad():
        tf = ao.teacher_features(tsd, img_o, lab_lr)
    assert torch.allclose(img_dev.cpu(), img_o, atol=1e-5)  # z-scored in place, then fed to the student
    osd = {k: v.clone().requires_grad_() for k, v in ssd.items() if k in so.segmodel_shapes(PLAN)}
    seg_lr, seg_sr, skips = so.seg_model(osd, img_o, PLAN, return_features=True)
    ref = ao.robust_ce(seg_lr, lab_lr[:, 0], unc)
    p = torch.softmax(seg_sr, 1)[:, 1:]
    oh = (lab_hr == 1).float()
    dc = (2 * (p * oh).sum((2, 3, 4)) + 1e-5) / torch.clip(oh.sum((2, 3, 4)) + p.sum((2, 3, 4)) + 1e-5, 1e-8)
    ref = ref + ao.robust_ce(seg_sr, lab_hr[:, 0], None) - dc.mean()
    ref = ref + ao.distiller_loss(dsd["distill.weight"], dsd["distill.bias"], skips[1], tf[1], 0.0, 1.0, 1.0)
    assert abs(loss.item() - ref.item()) <= 2e-4 * abs(ref.item()), (loss.item(), ref.item())
    # ... and the optimizer step that follows (train_all.py:554-556): the same SGD on the oracle's parameters must move
    # every tensor the same way.  delta = new - old isolates the gradient (momentum buffer = gradient on the first step).
    def oracle_step(dt):
        dw = dsd["distill.weight"].to(dt).clone().requires_grad_()
        db = dsd["distill.bias"].to(dt).clone().requires_grad_()
        o2 = {k: v.detach().to(dt).clone().requires_grad_() for k, v in osd.items()}
        s_lr, s_sr, sk = so.seg_model(o2, img_o.to(dt), PLAN, return_features=True)
        p_ = torch.softmax(s_sr, 1)[:, 1:]
        ohd = oh.to(dt)
        dc_ = (2 * (p_ * ohd).sum((2, 3, 4)) + 1e-5) / torch.clip(ohd.sum((2, 3, 4)) + p_.sum((2, 3, 4)) + 1e-5, 1e-8)
        r2 = ao.robust_ce(s_lr, lab_lr[:, 0], unc.to(dt)) + ao.robust_ce(s_sr, lab_hr[:, 0], None) - dc_.mean() + \
            ao.distiller_loss(dw, db, sk[1], tf[1].to(dt), 0.0, 1.0, 1.0)
        olds = {k: v.detach().clone() for k, v in o2.items()}
        oopt = torch.optim.SGD(list(o2.values()) + [dw, db], lr=1e-3, momentum=0.99, nesterov=True, weight_decay=3e-5)
        r2.backward()
        oopt.step()
        return {k: (v.detach() - olds[k]).double() for k, v in o2.items()}, (dw.detach() - dsd["distill.weight"].to(dt)).double()

    d32, dd32 = oracle_step(torch.float32)
    d64, _ = oracle_step(torch.float64)
    new = {canonical(k): v.detach().cpu() for k, v in student.state_dict().items()}
    worst = ("", 0.0, 0.0)
    for k in d32:
        if k.endswith("conv.bias"):       # zero gradient behind InstanceNorm: the update is weight decay only
            continue
        d_hip = (new[k] - ssd[k]).double()
        e = float((d_hip - d32[k]).norm() / (d32[k].norm() + 1e-30))
        cond = float((d32[k] - d64[k]).norm() / (d64[k].norm() + 1e-30))   # the oracle's own fp32 rounding distance
        # small InstanceNorm groups (192 voxels at the deepest stage) make some of these sums cancel: calibrated bar
        assert e <= max(2e-3, 6 * cond), (k, e, cond)
        if e > worst[1]:
            worst = (k, e, cond)
    e_d = float(((dist.distill.weight.detach().cpu() - dsd["distill.weight"]).double() - dd32).norm() / dd32.norm())
    print("joint step: worst parameter-update l2-rel vs the oracle's SGD step (name, hip-vs-fp32, fp32-vs-fp64):", worst,
          "distiller", e_d)
    assert e_d <= 2e-3, e_d


def test_teacher_stem_shared_between_windows_gpu():
    """encoder_on_windows (per-slice stem responses + window assembly) against the encoder run on the explicit window
    batch with UNet_3D_3D.forward's per-window mean subtraction (ref train_all.py:85-112, FLAVR_arch.py:181)."""
    import torch.nn.functional as F
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    from rehrseg_amd.models.FLAVR.resnet_3D import encoder_on_windows
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    enc = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True).to(dev).eval().encoder
    B, D, H, W = 2, 9, 64, 48
    x = torch.randn(B, 2, D, H, W, device=dev) * 2 + 0.7
    padded = F.pad(x, (0, 0, 0, 0, 1, 2))
    with torch.no_grad():
        got = encoder_on_windows(enc, padded[:, :, :D + 2], D - 1, 1)
        win = padded.unfold(2, 4, 1)[:, :, :D - 1].permute(0, 2, 1, 5, 3, 4).reshape(B * (D - 1), 2, 4, H, W).contiguous()
        win[:, 0:1] = win[:, 0:1] - win[:, 0:1].mean(dim=(2, 3, 4), keepdim=True)
        ref = enc(win, upto=1)
    for a, e in zip(got, ref):
        assert a.shape == e.shape
        err = ((a - e).abs().max() / e.abs().max()).item()
        assert err <= 2e-5, err


def test_teacher_on_its_own_stream_changes_nothing():
    """train_segsr_step(teacher_stream=True) runs the frozen teacher's pass on a second HIP stream next to the student's
    forward.  Same kernels on the same operands: the loss and every gradient of the step equal the
    single-stream step's up to the summation order of the fp64 statistics atomics.  A missing join or a recycled buffer
    would show as O(1) errors (the shared zero-accumulator pool did, as NaN, before it was keyed per stream)."""
    import itertools
    dev = torch.device("cuda:0")
    teacher, _ = _flavr(True, dev)
    teacher.eval()
    img = det_input("st2.img", (2, 1, 6, 32, 32), "rand") * 2 + 0.5
    lab_lr = det_input("st2.lr", (2, 1, 6, 32, 32), "randint2").to(dev)
    lab_hr = det_input("st2.hr", (2, 1, 24, 32, 32), "randint2").to(dev)
    unc = (1 - det_input("st2.u", (2, 1, 6, 32, 32), "rand") * 0.99).to(dev)
    out = {}
    for tag, side in (("single", False), ("side", True), ("again", False)):
        student, _ = build(PLAN, dev)
        dist = Distiller(64, 64, 0.0, 1.0, 1.0)
        dist.load_state_dict({k: det_tensor(k, tuple(v.shape)) for k, v in dist.state_dict().items()})
        dist = dist.to(dev)
        params = list(itertools.chain(student.parameters(), dist.parameters()))
        opt = torch.optim.SGD(params, lr=0.0)                 # lr 0: the step leaves the gradients to look at
        loss = train_segsr_step(student, teacher, dist, opt, img.clone().to(dev), lab_lr, lab_hr, unc,
                                su._build_loss(False, weight_dice=0), su._build_loss(False, weight_dice=1),
                                teacher_stream=side)
        torch.cuda.synchronize()
        out[tag] = (float(loss), [p.grad.detach().clone() if p.grad is not None else None for p in params])

    def worst(a, b):
        return max(float((x - y).abs().max()) / (float(x.abs().max()) + 1e-30) for x, y in zip(a, b) if x is not None)
    repro = worst(out["single"][1], out["again"][1])
    d = worst(out["single"][1], out["side"][1])
    print("teacher stream vs single stream: loss", out["side"][0], out["single"][0], "gradients", d, "run-to-run", repro)
    assert abs(out["side"][0] - out["single"][0]) <= 1e-6 * abs(out["single"][0])
    assert d <= max(2e-6, 4 * repro), (d, repro)
