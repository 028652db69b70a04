"""G8 on the MI355X: the HIP SegModel (and the fused DC+CE loss kernel) against the fixtures captured from
the reference's own SegModel.forward / MyUnetDecoder.forward / DC_and_weighted_CE_loss.forward
(tools/gen_golden_segmodel.py; bases = eager-torch stand-ins, "parity unpinned").
Tolerances: forward 1e-4 of the tensor's max, losses 1e-5, gradients 1e-3 (north_star's bar) -- the fixture is
the reference's fp32 CPU run."""
import numpy as np
import pytest
import torch

from rehrseg_amd.utils import seg_utils as su
from test_segmodel_cpu import build, canonical
from test_segmodel_golden_cpu import CASES, check_against_fixture, fixture, loss_cases, relmax, stage2_loss

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["small", "aniso4"])
def test_hip_segmodel_against_reference_segmodel(tag):
    dev = torch.device("cuda:0")
    cfg, G = CASES[tag], fixture(tag)
    m, _ = build(cfg, dev)
    out, out_up, skips = m(torch.from_numpy(G["x"]).to(dev), return_inetermediate_feature=True)
    loss, l_lr, l_hr = stage2_loss(out, out_up, skips[1], G, tag, su._build_loss(), dev)   # fused loss kernel
    loss.backward()
    grads = {canonical(k): p.grad for k, p in m.named_parameters()}
    worst = check_against_fixture(tag, out, out_up, skips, (loss, l_lr, l_hr), grads, 1e-4, 1e-3)
    print(f"{tag}: worst full-gradient l2-rel vs the reference {worst[1]:.2e} ({worst[0]})")
    mds, _ = build(cfg, dev, deep_supervision=True)
    with torch.no_grad():
        outs, _ = mds(torch.from_numpy(G["x"]).to(dev))
    for i, o in enumerate(outs):
        assert relmax(o.cpu(), G[f"ds_out{i}"]) < 1e-4


def test_fused_dc_ce_kernel_against_reference_loss():
    dev = torch.device("cuda:0")
    for tag, wd, lg, tg, un, ref, rgrad in loss_cases():
        x = torch.from_numpy(lg).to(dev).requires_grad_()
        v = su._build_loss(weight_dice=wd)(x, torch.from_numpy(tg).to(dev), None if un is None else torch.from_numpy(un).to(dev))
        v.backward()
        assert abs(float(v.detach()) - ref) <= 1e-5 * max(1.0, abs(ref)), (tag, float(v.detach()), ref)
        assert relmax(x.grad.cpu(), rgrad) < 1e-4, tag
