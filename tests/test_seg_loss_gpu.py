"""Fused segmentation loss (rehr_seg_loss_{fwd,bwd}_f32) against the torch composition of the same loss
(rehrseg_amd/utils/seg_utils.py, itself checked against the reference's loss fixtures in
tests/test_aux_cpu.py) evaluated on the CPU in fp64.  Tolerance 1e-5 relative for the value, 1e-4 of the
largest gradient magnitude for the logit gradient."""
import pytest
import torch

from rehrseg_amd.utils import seg_utils as su

pytestmark = pytest.mark.gpu


def _ref(loss_mod, logits, target, unc):
    """The torch composition, forced (CPU tensors never take the fused path)."""
    lg = logits.detach().double().cpu().requires_grad_(True)
    val = loss_mod.double()(lg, target.cpu().double(), None if unc is None else unc.cpu().double())
    val.backward()
    return val.detach(), lg.grad


@pytest.mark.parametrize("C", [2, 3])
@pytest.mark.parametrize("with_unc", [False, True])
@pytest.mark.parametrize("weight_dice", [1.0, 0.5])
def test_fused_dc_ce_matches_composition(C, with_unc, weight_dice):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11 + C)
    N, D, H, W = 2, 6, 10, 12
    logits = (torch.randn(N, C, D, H, W, generator=g) * 2.0)
    target = torch.randint(0, C, (N, 1, D, H, W), generator=g).float()
    unc = (1.0 - torch.randint(0, 256, (N, 1, D, H, W), generator=g).float() / 255.0 * 0.99) if with_unc else None
    mod = su._build_loss(weight_dice=weight_dice)
    rv, rg = _ref(su._build_loss(weight_dice=weight_dice), logits, target, unc)

    lg = logits.to(dev).requires_grad_(True)
    val = mod(lg, target.to(dev), None if unc is None else unc.to(dev))
    (val * 3.0).backward()   # a non-unit upstream gradient goes through the device scalar
    assert abs(float(val.detach()) - float(rv)) <= 1e-5 * max(1.0, abs(float(rv)))
    got = lg.grad.double().cpu() / 3.0
    scale = float(rg.abs().max())
    assert float((got - rg).abs().max()) <= 1e-4 * scale


def test_fused_loss_full_size_properties():
    """cfg-3 HR-logit size: value finite, gradient sums to zero over classes at every voxel (softmax),
    and a uniform shift of the logits leaves the loss unchanged."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    N, C, D, H, W = 2, 2, 128, 128, 128
    logits = torch.randn(N, C, D, H, W, generator=g).to(dev).requires_grad_(True)
    target = torch.randint(0, C, (N, 1, D, H, W), generator=g).float().to(dev)
    mod = su._build_loss()
    val = mod(logits, target)
    val.backward()
    assert torch.isfinite(val)
    assert float(logits.grad.sum(1).abs().max()) <= 1e-9
    val2 = mod((logits.detach() + 7.5), target)
    assert abs(float(val2) - float(val.detach())) <= 1e-5


def test_fused_bce_dice_against_reference_fixture_and_composition():
    """BCEDiceLoss on the device (rehr_bce_dice_{fwd,bwd}_f32) against the value / gradient the REFERENCE's BCEDiceLoss
    produced (tests/golden/aux_losses_teacher.npz, tools/gen_golden_losses.py) and, on a two-channel batch with a
    non-contiguous logits slice, against the torch composition in fp64."""
    import os
    import numpy as np
    from oracle.detinit import det_input
    dev = torch.device("cuda:0")
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "aux_losses_teacher.npz"))
    x = det_input("bcedice.x", (2, 1, 4, 16, 16)).to(dev).requires_grad_()
    t = det_input("bcedice.t", (2, 1, 4, 16, 16), "randint2").to(dev)
    l = su.BCEDiceLoss(1.0, 1.0)(x, t)
    l.backward()
    assert abs(float(l.detach()) - float(G["bcedice_loss"])) < 1e-6
    assert float((x.grad.cpu() - torch.from_numpy(G["bcedice_grad"])).abs().max()) <= 1e-4 * float(np.abs(G["bcedice_grad"]).max())
    g = torch.Generator().manual_seed(3)
    full = (torch.randn(3, 4, 4, 24, 20, generator=g) * 2).to(dev).requires_grad_()
    tgt = torch.randint(0, 2, (3, 2, 4, 24, 20), generator=g).float().to(dev)
    val = su.BCEDiceLoss(0.7, 1.3)(full[:, 1:3], tgt)          # a channel slice, like hat[:, 1:] in train_sr
    (val * 2.0).backward()
    ref_in = full.detach().double().cpu().requires_grad_()
    ref = su.BCEDiceLoss(0.7, 1.3).double()(ref_in[:, 1:3], tgt.double().cpu())
    (ref * 2.0).backward()
    assert abs(float(val.detach()) - float(ref.detach())) <= 1e-6 * abs(float(ref.detach()))
    assert float((full.grad.double().cpu() - ref_in.grad).abs().max()) <= 1e-5 * float(ref_in.grad.abs().max())
