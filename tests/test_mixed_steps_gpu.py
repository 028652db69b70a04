"""BASELINE.json configs[4] under a checker: the joint stage-2 step (FLAVR teacher + SegModel student + Distiller),
the FLAVR U-Net and single fused blocks under `ops.mixed_precision()` against the CPU oracles.

How bf16 ROUNDING is told from kernel ERROR: the oracles take `emu=Bf16Emu()`
(oracle/bf16_emul.py) and then round -- in fp64 arithmetic -- at exactly the places where the device path stores
bf16 (operands of the matrix-core layers, conv outputs in front of InstanceNorm / SEGating, block outputs,
activation gradients).  For every compared tensor three distances are formed:

  e_hip32 = |HIP(bf16) - oracle(fp32)|      what a user sees against the reference's fp32 run
  e_emu32 = |oracle(bf16-emulated, fp64) - oracle(fp32)|   what bf16 rounding alone does to that tensor
  e_hipE  = |HIP(bf16) - oracle(bf16-emulated)|            what is left: accumulation order, elements that sit on a
                                                            bf16 rounding boundary, and any kernel error

Bars: e_hip32 <= max(floor, 2.5 * e_emu32) per tensor -- the device's deviation from fp32 is of the size rounding alone
produces -- and e_hipE <= max(floor, 1.5 * e_emu32).  Where the emulation can follow the device (shallow layers, forward
tensors) e_hipE comes out several times SMALLER than e_emu32 (printed); in the deep layers of a randomly initialised
stack the gradient's rounding noise is chaotic (which way an element on a bf16 boundary rounds decides an activation
mask downstream), two realisations of the same rounding model then sit sqrt(2) noise magnitudes apart, hence 1.5.  A
wrong tap, scale or tile in a bf16 kernel moves e_hipE and e_hip32 to O(1) while e_emu32 stays where it is.
"""
import itertools

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import aux_oracle as ao
from oracle import flavr_oracle as fo
from oracle import segmodel_oracle as so
from oracle.bf16_emul import Bf16Emu
from oracle.detinit import det_input, det_tensor
from rehrseg_amd import ops
from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
from rehrseg_amd.models.seg_model import Distiller
from rehrseg_amd.train_steps import get_intermediate_features, train_segsr_step
from rehrseg_amd.utils import seg_utils as su
from test_segmodel_cpu import build, canonical

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PLAN = dict(n_stages=3, features_per_stage=[32, 64, 96], kernel_sizes=[[1, 3, 3], [1, 3, 3], [3, 3, 3]],
            strides=[[1, 1, 1], [1, 2, 2], [2, 2, 2]], n_conv_per_stage=[2, 2, 2], n_conv_per_stage_decoder=[2, 2],
            num_classes=2, upscale=4)


def l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def three_way(name, hip, o32, oemu, floor32, floorE, rows):
    e_hip32, e_emu32, e_hipE = l2rel(hip, o32), l2rel(oemu, o32), l2rel(hip, oemu)
    rows.append((name, e_hip32, e_emu32, e_hipE))
    assert e_hip32 <= max(floor32, 2.5 * e_emu32), (name, e_hip32, e_emu32, e_hipE)
    assert e_hipE <= max(floorE, 1.5 * e_emu32), (name, e_hip32, e_emu32, e_hipE)


def show(tag, rows, top=10):
    print(f"[{tag}] {'tensor':58s} hip-vs-fp32   emu-vs-fp32   hip-vs-emu")
    for n, a, b, c in sorted(rows, key=lambda r: -r[1])[:top]:
        print(f"[{tag}] {n:58s} {a:.3e}    {b:.3e}    {c:.3e}")


def _joint_oracle(plan, tsd, ssd, dsd, img, lab_lr, lab_hr, unc, dt, emu):
    """train_all.py:531-555 composed from the oracles in dtype `dt`; returns loss, teacher level-1 features, the
    student's LR/HR logits and the gradient of every student / distiller parameter."""
    img_o = img.clone().to(dt)
    t = {k: v.to(dt) for k, v in tsd.items()}
    with torch.no_grad():
        tf = ao.teacher_features(t, img_o, lab_lr.to(dt), emu=emu, upto=1)
    o = {k: v.detach().to(dt).clone().requires_grad_() for k, v in ssd.items() if k in so.segmodel_shapes(plan)}
    dw = dsd["distill.weight"].to(dt).clone().requires_grad_()
    db = dsd["distill.bias"].to(dt).clone().requires_grad_()
    s_lr, s_sr, sk = so.seg_model(o, img_o, plan, return_features=True, emu=emu)
    loss = ao.dc_and_weighted_ce(s_lr, lab_lr.to(dt), unc.to(dt), weight_dice=0.0) + \
        ao.dc_and_weighted_ce(s_sr, lab_hr.to(dt), None) + \
        ao.distiller_loss(dw, db, sk[1], tf[1], 0.0, 1.0, 1.0, emu=emu)
    loss.backward()
    grads = {k: v.grad.double() for k, v in o.items() if v.grad is not None}   # (unused deep-supervision heads: None)
    grads["distill.weight"], grads["distill.bias"] = dw.grad.double(), db.grad.double()
    return float(loss.detach()), tf[1].double(), s_lr.detach().double(), s_sr.detach().double(), grads


def _joint_step_three_way(tag, plan, shape, floors):
    """train_segsr_step under mixed_precision() against the oracle step in fp32 and in the bf16-emulating fp64 form:
    loss, teacher features, logits and the gradient the step's backward leaves in every parameter (lr = 0 keeps the
    weights; the gradients of a step, not fp32-quantised parameter differences, are what is compared)."""
    dev = torch.device(DEV)
    B, D, H, W = shape
    teacher = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True)
    tsd = {k: det_tensor(k, tuple(v.shape)) for k, v in teacher.state_dict().items()}
    teacher.load_state_dict(tsd)
    teacher = teacher.to(dev).eval()
    student, ssd = build(plan, dev)
    dist = Distiller(64, 64, 0.0, 1.0, 1.0)
    dsd = {k: det_tensor(k, tuple(v.shape)) for k, v in dist.state_dict().items()}
    dist.load_state_dict(dsd)
    dist = dist.to(dev)
    img = det_input(tag + ".img", (B, 1, D, H, W), "rand") * 2 + 0.5
    lab_lr = det_input(tag + ".lr", (B, 1, D, H, W), "randint2")
    lab_hr = det_input(tag + ".hr", (B, 1, 4 * D, H, W), "randint2")
    unc = 1 - det_input(tag + ".u", (B, 1, D, H, W), "rand") * 0.99
    opt = torch.optim.SGD(itertools.chain(student.parameters(), dist.parameters()), lr=0.0)
    with ops.mixed_precision():
        with torch.no_grad():
            imz = img.clone().to(dev)
            tf_hip = get_intermediate_features(teacher, imz, lab_lr.to(dev), dev, levels=(1,))[1]
            s_lr_hip, s_sr_hip = student(imz)                    # z-scored in place by the teacher pass
        assert tf_hip.dtype == torch.bfloat16                    # the teacher really ran on the bf16 kernels
        loss = train_segsr_step(student, teacher, dist, opt, img.clone().to(dev), lab_lr.to(dev), lab_hr.to(dev),
                                unc.to(dev), su._build_loss(False, weight_dice=0), su._build_loss(False, weight_dice=1))
    l32, tf32, lr32, sr32, g32 = _joint_oracle(plan, tsd, ssd, dsd, img, lab_lr, lab_hr, unc, torch.float32, None)
    lE, tfE, lrE, srE, gE = _joint_oracle(plan, tsd, ssd, dsd, img, lab_lr, lab_hr, unc, torch.float64, Bf16Emu())
    rows = []
    three_way("teacher level-1 features", tf_hip.float(), tf32, tfE, 1e-2, 2e-3, rows)
    three_way("student LR logits", s_lr_hip, lr32, lrE, 2e-2, 5e-3, rows)
    three_way("student HR logits", s_sr_hip, sr32, srE, 2e-2, 5e-3, rows)
    hip = {canonical(k): p.grad for k, p in student.named_parameters() if p.grad is not None}
    hip["distill.weight"], hip["distill.bias"] = dist.distill.weight.grad, dist.distill.bias.grad
    for k in g32:
        if k.endswith("conv.bias") and not k.startswith("sr_head"):   # in front of InstanceNorm: identically zero gradient
            continue
        three_way("grad " + k, hip[k], g32[k], gE[k], *floors, rows)
    show(tag, rows, 16)
    print(f"[{tag}] loss hip", float(loss), "oracle fp32", l32, "oracle bf16-emulated", lE)
    assert abs(float(loss) - l32) <= max(2e-3 * abs(l32), 2.5 * abs(lE - l32)), (float(loss), l32, lE)
    assert abs(float(loss) - lE) <= max(1e-3 * abs(lE), 1.5 * abs(lE - l32)), (float(loss), l32, lE)
    return rows


def test_joint_step_mixed_precision_toy_plan():
    """The first time the FLAVR teacher and the Distiller run in bf16 under a checker (3-stage plan, 2 x 1 x 6 x 32 x 32)."""
    _joint_step_three_way("joint bf16 toy", PLAN, (2, 6, 32, 32), (2e-2, 1e-2))


def test_joint_step_mixed_precision_full_depth_plan():
    """The benchmarked six-stage anisotropic plan (BASELINE configs[3] / [4]: 22 conv layers, strides down to 1/32) at
    1 x 1 x 32 x 64 x 64, where the bf16-emulating fp64 oracle still runs in seconds.  This is the calibration the
    full-size cfg-5 test (tests/test_joint_step_full_size_gpu.py) leans on: how far bf16 ROUNDING ALONE moves the deep
    gradients of a randomly initialised InstanceNorm stack (e_emu32, tens of percent at the bottom stages), and that
    the device path stays closer to the rounding-emulated oracle than that oracle is to fp32."""
    rows = _joint_step_three_way("joint bf16 full-depth", so.ANISO_PLAN, (1, 32, 64, 64), (3e-2, 2e-2))
    deep = [r for r in rows if "encoder.stages.4" in r[0] or "encoder.stages.5" in r[0]]
    print("[joint bf16 full-depth] deepest stages: median emu-vs-fp32", sorted(r[2] for r in deep)[len(deep) // 2],
          "median hip-vs-emu", sorted(r[3] for r in deep)[len(deep) // 2])


def test_flavr_mixed_precision_against_reference_fixture():
    """UNet_3D_3D(2,'unet_18',4,4) forward + every parameter gradient in bf16 mixed precision against the REFERENCE's
    fp32 run (tests/golden/flavr_c2_n4.npz), with the oracle (pinned by that same fixture) as fp32 / bf16-emulated
    go-between for the per-tensor calibration."""
    import os
    from test_flavr_model_cpu import GOLD, build as build_flavr
    g = np.load(os.path.join(GOLD, "flavr_c2_n4.npz"))
    m, _ = build_flavr(g, DEV)
    x, tgt = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])
    with ops.mixed_precision():
        out = m(x.clone().to(DEV))
        loss = (out.float() - tgt.to(DEV)).abs().mean()
        loss.backward()
    sd = {k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()}
    res = {}
    for tag, dt, emu in (("32", torch.float32, None), ("E", torch.float64, Bf16Emu())):
        osd = {k: v.to(dt).clone().requires_grad_() for k, v in sd.items()}
        o = fo.unet_3d_3d(osd, x.clone().to(dt), 2, 4, 4, emu=emu)
        ol = (o - tgt.to(dt)).abs().mean()
        ol.backward()
        res[tag] = (o.detach(), float(ol.detach()), {k: v.grad for k, v in osd.items()})
    assert float((res["32"][0] - torch.from_numpy(g["out"])).abs().max()) <= 1e-4 * float(np.abs(g["out"]).max())  # oracle == reference
    rows = []
    three_way("output", out.float(), res["32"][0], res["E"][0], 2e-2, 5e-3, rows)
    three_way("output vs the reference fixture itself", out.float(), torch.from_numpy(g["out"]), res["E"][0], 2e-2, 5e-3, rows)
    for k, p in m.named_parameters():
        g32, gE = res["32"][2][k], res["E"][2][k]
        if p.grad is None or g32 is None or float(g32.norm()) == 0.0:
            continue
        three_way("grad " + k, p.grad, g32, gE, 3e-2, 1.5e-2, rows)
    show("flavr bf16", rows, 14)
    l32, lE = res["32"][1], res["E"][1]
    print("[flavr bf16] loss hip", float(loss), "reference", float(g["loss"]), "oracle bf16-emulated", lE)
    assert abs(float(loss) - float(g["loss"])) <= max(2e-3 * abs(l32), 2.5 * abs(lE - l32))


@pytest.mark.parametrize("case", ["in_3x3x3", "in_stride2", "in_1x3x3_stride122", "tconv_2x2x2", "tconv_1x2x2", "se_res", "se_tconv_344"])
def test_single_layer_bf16_gradients_against_fp64_on_the_same_rounded_operands(case):
    """one fused layer in bf16 against torch fp64 fed the SAME bf16-rounded input, weights and
    output gradient, with the device's interior rounding point (conv output stored as bf16 in front of the
    normalisation / gate) emulated.  What remains is accumulation order and boundary elements, so the bars are tight:
    forward 2^-7 of max (one bf16 ulp of the stored output), input gradient 1.5e-2 l2-rel (bf16 store + mask elements
    on a rounding boundary), weight / affine gradients 5e-3 (fp32 stores)."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(17)
    emu = Bf16Emu()
    q = lambda t: t.to(torch.bfloat16).float()
    if case.startswith("in_"):
        k, s = {"in_3x3x3": ((3, 3, 3), (1, 1, 1)), "in_stride2": ((3, 3, 3), (2, 2, 2)),
                "in_1x3x3_stride122": ((1, 3, 3), (1, 2, 2))}[case]
        ci, co = 64, 96
        x = q(torch.randn(2, ci, 8, 24, 20, generator=gen))
        w = (torch.randn(co, ci, *k, generator=gen) / (ci * k[0] * 9) ** 0.5)
        b = torch.randn(co, generator=gen) * 0.1
        p1, p2 = torch.rand(co, generator=gen) + 0.5, torch.randn(co, generator=gen) * 0.1
        pad = tuple((i - 1) // 2 for i in k)
        hip = lambda xx, ww, bb, a, c: ops.fused_conv3d(xx, ww, bb, s, pad, inorm=(a, c), act=ops.ACT_LRELU, slope=0.01)

        def ref(xx, ww, bb, a, c):
            y = emu.grad(F.conv3d(xx, emu.weight(ww), bb, s, pad))    # dz: summed in fp32, stored once as bf16
            mean, var = y.mean((2, 3, 4), keepdim=True), y.var((2, 3, 4), unbiased=False, keepdim=True)
            return F.leaky_relu((emu.fwd(y) - mean) * torch.rsqrt(var + 1e-5) * a.view(1, -1, 1, 1, 1) + c.view(1, -1, 1, 1, 1), 0.01)
    elif case.startswith("tconv_"):
        s = (2, 2, 2) if case == "tconv_2x2x2" else (1, 2, 2)
        ci, co = 128, 64
        x = q(torch.randn(2, ci, 4, 12, 10, generator=gen))
        w = torch.randn(ci, co, *s, generator=gen) / ci ** 0.5
        b = torch.randn(co, generator=gen) * 0.1
        p1 = p2 = None
        hip = lambda xx, ww, bb, a, c: ops.fused_conv3d(xx, ww, bb, s, 0, transposed=True)
        ref = lambda xx, ww, bb, a, c: F.conv_transpose3d(xx, emu.weight(ww), bb, s)
    else:
        tr = case == "se_tconv_344"
        ci, co = 64, 64
        x = q(torch.randn(2, ci, 4, 16, 12, generator=gen))
        w = (torch.randn(ci, co, 3, 4, 4, generator=gen) / (ci * 12) ** 0.5) if tr else \
            (torch.randn(co, ci, 3, 3, 3, generator=gen) / (ci * 27) ** 0.5)
        b = torch.randn(co, generator=gen) * 0.1
        p1, p2 = torch.randn(co, co, 1, 1, 1, generator=gen) * 0.3, torch.randn(co, generator=gen) * 0.1
        res_t = None if tr else q(torch.randn(2, co, 4, 16, 12, generator=gen))
        if tr:
            hip = lambda xx, ww, bb, a, c: ops.fused_conv3d(xx, ww, bb, (1, 2, 2), (1, 1, 1), transposed=True, se=(a, c),
                                                            act=ops.ACT_LRELU, slope=0.2)
        else:
            hip = lambda xx, ww, bb, a, c: ops.fused_conv3d(xx, ww, bb, 1, 1, se=(a, c), res=res_t.to(DEV).to(torch.bfloat16),
                                                            act=ops.ACT_RELU)

        def ref(xx, ww, bb, a, c):
            y = F.conv_transpose3d(xx, emu.weight(ww), bb, (1, 2, 2), (1, 1, 1)) if tr else F.conv3d(xx, emu.weight(ww), bb, 1, 1)
            y = emu.grad(y)                                           # the gate path's constant is added to the stored bf16 dz
            gate = torch.sigmoid(F.conv3d(y.mean((2, 3, 4), keepdim=True), a, c))
            z = emu.act(y) * gate
            return F.leaky_relu(z, 0.2) if tr else torch.relu(z + res_t.to(xx.dtype))
    # device run
    leaves = [t.clone().to(DEV).requires_grad_() if t is not None else None for t in (x, w, b, p1, p2)]
    with ops.mixed_precision():
        y_hip = hip(*leaves)
    assert y_hip.dtype == torch.bfloat16
    dy = q(torch.randn(y_hip.shape, generator=gen))
    y_hip.backward(dy.to(DEV).to(torch.bfloat16))
    # fp64 run on the same rounded operands
    rl = [t.clone().double().requires_grad_() if t is not None else None for t in (x, w, b, p1, p2)]
    y_ref = ref(*rl)
    y_ref.backward(dy.double())
    fwd = float((y_hip.float().cpu().double() - y_ref.detach()).abs().max() / y_ref.detach().abs().max())
    errs = {n: l2rel(a.grad, r.grad) for n, a, r in zip(("dx", "dw", "db", "dp1", "dp2"), leaves, rl)
            if a is not None and r.grad is not None and float(r.grad.norm()) > 0}
    print(f"[layer bf16 {case}] fwd max-rel {fwd:.2e} " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert fwd <= 2.0 ** -7
    assert errs["dx"] <= 1.5e-2
    assert errs["dw"] <= 5e-3
    for k in ("db", "dp1", "dp2"):
        if k in errs and not (case.startswith("in_") and k == "db"):      # conv bias in front of InstanceNorm: zero gradient
            assert errs[k] <= 5e-3, (k, errs[k])
