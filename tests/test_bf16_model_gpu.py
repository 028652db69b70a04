"""Mixed precision (BASELINE.json configs[4]) end to end: the fused blocks and the SegModel under
`ops.mixed_precision()` against the fp32 references.

Tolerances (bf16 has 8 significant bits, eps = 2^-8 = 3.9e-3; every layer rounds its activations once, fp32
accumulation and fp32/fp64 statistics inside): fused block forward 2e-2 of the tensor's max, block gradients 6e-2
l2-relative (activation masks flip where the pre-activation is within the bf16 rounding of zero); SegModel (10 convs deep at the small plan) logits 5e-2 of max, loss 2e-2 relative, parameter gradients
0.25 l2-relative (0.17 observed on the first conv, the deepest gradient) against the REFERENCE's fp32 run (tests/golden/segmodel_small.npz).  Observed values are printed."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from rehrseg_amd import hip_backend, ops
from rehrseg_amd.utils import seg_utils as su
from test_segmodel_cpu import build, canonical
from test_segmodel_golden_cpu import CASES, fixture, stage2_loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def l2rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30))


def relmax(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().cpu().abs().max() + 1e-30))


def _params(gen, *shapes):
    return [torch.randn(s, generator=gen).to(DEV).requires_grad_() for s in shapes]


@pytest.mark.parametrize("mode", ["in", "se", "plain"])
def test_fused_block_bf16_vs_fp32_path(mode):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 6, 20, 24, generator=g).to(DEV)
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) / (64 * 27) ** 0.5).to(DEV).requires_grad_()
    b = (torch.randn(64, generator=g) * 0.1).to(DEV).requires_grad_()
    res = torch.randn(2, 64, 6, 20, 24, generator=g).to(DEV)
    proj = torch.randn(2, 64, 6, 20, 24, generator=g).to(DEV)
    if mode == "in":
        p1, p2 = (torch.rand(64, generator=g) + 0.5).to(DEV).requires_grad_(), (torch.randn(64, generator=g) * 0.1).to(DEV).requires_grad_()
        kw = dict(inorm=(p1, p2), act=ops.ACT_LRELU, slope=0.01)
    elif mode == "se":
        p1, p2 = (torch.randn(64, 64, 1, 1, 1, generator=g) * 0.3).to(DEV).requires_grad_(), torch.zeros(64, device=DEV).requires_grad_()
        kw = dict(se=(p1, p2), res=None, act=ops.ACT_RELU)
    else:
        p1 = p2 = None
        kw = dict(act=ops.ACT_RELU)
    outs = {}
    for mixed in (False, True):
        xin = x.clone().requires_grad_()
        rin = res.clone().requires_grad_()
        for t in (w, b, p1, p2):
            if t is not None:
                t.grad = None
        k = dict(kw)
        if mode == "se":
            k["res"] = rin
        before = hip_backend.wino_launches
        with ops.mixed_precision(mixed):
            y = ops.fused_conv3d(xin, w, b, 1, 1, **k)
            (y.float() * proj).mean().backward()
        assert y.dtype == (torch.bfloat16 if mixed else torch.float32)
        if mixed:
            assert hip_backend.wino_launches == before     # the bf16 path has no fp32 Winograd launch in it
        outs[mixed] = (y.detach().float(), xin.grad.float(), w.grad.clone(), b.grad.clone(),
                       None if p1 is None else p1.grad.clone(), rin.grad.float() if mode == "se" else None)
    y32, dx32, dw32, db32, dp32, dr32 = outs[False]
    y16, dx16, dw16, db16, dp16, dr16 = outs[True]
    print(mode, "fwd", relmax(y16, y32), "dx", l2rel(dx16, dx32), "dw", l2rel(dw16, dw32), "db", l2rel(db16, db32))
    assert relmax(y16, y32) < 2e-2
    # 0.038-0.040 observed: the activation masks of the two runs differ wherever |pre-activation| is below the bf16
    # rounding (~0.3 % of the elements), and a flipped mask element is a 100 % error of that element
    assert l2rel(dx16, dx32) < 6e-2 and l2rel(dw16, dw32) < 6e-2
    if mode != "in":          # behind InstanceNorm the conv bias gradient is identically zero
        assert l2rel(db16, db32) < 6e-2
    if dp32 is not None:
        assert l2rel(dp16, dp32) < 8e-2
    if dr32 is not None:
        assert l2rel(dr16, dr32) < 6e-2


def test_segmodel_mixed_precision_against_reference_fixture():
    tag = "small"
    cfg, G = CASES[tag], fixture(tag)
    m, _ = build(cfg, DEV)
    with ops.mixed_precision():
        out, out_up, skips = m(torch.from_numpy(G["x"]).to(DEV), return_inetermediate_feature=True)
        assert out.dtype == torch.float32 and out_up.dtype == torch.float32     # logits: fp32 thin heads
        assert skips[1].dtype == torch.bfloat16
        loss, l_lr, l_hr = stage2_loss(out, out_up, skips[1].float(), G, tag, su._build_loss(), DEV)
        loss.backward()
    fwd = max(relmax(out, torch.from_numpy(G["out"])), relmax(out_up, torch.from_numpy(G["out_up"])))
    lrel = abs(float(loss.detach()) - float(G["loss"])) / abs(float(G["loss"]))
    grads = {canonical(k): p.grad for k, p in m.named_parameters()}
    worst = ("", 0.0)
    for name in ("sr_head.2.bias", "sr_head.0.bias", "decoder.seg_layers.1.weight", "decoder.transpconvs.0.bias",
                 "encoder.stages.0.0.convs.0.conv.weight", "encoder.stages.1.0.convs.1.norm.weight",
                 "decoder.stages.1.convs.0.norm.bias"):
        e = l2rel(grads[name], torch.from_numpy(G["grad:" + name]))
        worst = max(worst, (name, e), key=lambda t: t[1])
    norm_err = max(abs(float(grads[str(n)].double().norm()) - r) / max(r, 1e-12) for n, r in zip(G["grad_names"], G["grad_norms"])
                   if not ("conv.bias" in str(n) and "sr_head" not in str(n)))
    print("segmodel mixed precision: fwd", fwd, "loss rel", lrel, "worst full grad", worst, "worst grad-norm rel", norm_err)
    assert fwd < 5e-2 and lrel < 2e-2 and worst[1] < 0.25 and norm_err < 0.15
