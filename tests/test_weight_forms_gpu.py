"""Kept weight forms (rehrseg_amd.hip_backend, "weight forms"): the packed panels and Winograd-domain weights of
nn.Parameters live across launches and are rebuilt when the parameter's version counter moves.  Keeping them must
change nothing: frozen networks and no-grad passes stop rebuilding and give the same bits, optimizer steps and in-place
updates are seen, dead parameters release their forms."""
import gc

import pytest
import torch

from oracle.detinit import det_tensor
from rehrseg_amd import hip_backend as be
from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
from rehrseg_amd.models.seg_model import SegModel

pytestmark = pytest.mark.gpu


def _flavr(dev):
    m = UNet_3D_3D(2, "unet_18", 4, 4)
    m.load_state_dict({k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()})
    return m.to(dev)


def _seg(dev):
    import torch.nn as nn
    torch.manual_seed(3)      # four stages with a two-source decoder: the halves of the virtual concat are forms too
    return SegModel(input_channels=1, num_classes=2, n_stages=4, upscale=4, features_per_stage=[32, 64, 128, 256],
                    conv_op=nn.Conv3d, kernel_sizes=[[3, 3, 3]] * 4, strides=[[1, 1, 1]] + [[2, 2, 2]] * 3,
                    n_conv_per_stage=[2] * 4, n_conv_per_stage_decoder=[2] * 3, conv_bias=True,
                    norm_op=nn.InstanceNorm3d, norm_op_kwargs={"eps": 1e-5, "affine": True}, dropout_op=None,
                    dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs={"inplace": True},
                    deep_supervision=False).to(dev)


def _first(y):
    return y[0] if isinstance(y, (tuple, list)) else y


@pytest.mark.parametrize("which", ["flavr", "seg"])
def test_validation_passes_between_fused_optimizer_steps_see_the_new_weights(which):
    """Fused optimizers do not move the parameters' version counters: the optimizer-step hook of hip_backend is what
    tells the kept forms.  Training steps (recorded passes: nothing kept) alternate with no-grad validation passes
    (forms kept); every validation output must equal the one computed without kept forms, bit for bit, and the
    recorded passes must not have touched the kept forms."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    if which == "flavr":
        x = torch.rand(2, 2, 4, 32, 32, generator=g).to(dev)
        m = _flavr(dev)
    else:
        x = torch.rand(1, 1, 32, 32, 32, generator=g).to(dev)
        m = _seg(dev)
    be.invalidate_weight_forms()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    per_pass = None
    for step in range(3):
        r0 = be.form_rebuilds
        opt.zero_grad()
        _first(m(x.clone())).abs().mean().backward()
        assert be.form_rebuilds == r0                      # a recorded pass over trainable weights keeps nothing
        opt.step()
        with torch.no_grad():
            y = _first(m(x.clone()))
            made = be.form_rebuilds - r0
            assert made > 10 and (per_pass is None or made == per_pass)    # every kept form once per update, no more
            per_pass = made
            y_again = _first(m(x.clone()))
            assert be.form_rebuilds - r0 == made           # a second pass on unchanged weights: nothing rebuilt
            be.WEIGHT_FORM_CACHE = False
            try:
                want = _first(m(x.clone()))
            finally:
                be.WEIGHT_FORM_CACHE = True
        assert torch.equal(y, want) and torch.equal(y_again, want), step
    be.invalidate_weight_forms()


def test_frozen_network_keeps_its_forms_and_sees_in_place_updates():
    dev = torch.device("cuda:0")
    be.invalidate_weight_forms()
    m = _flavr(dev).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    x = torch.rand(1, 2, 4, 32, 32, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        y0 = m(x.clone())
        n1 = be.form_rebuilds
        y1 = m(x.clone())
        assert be.form_rebuilds == n1                       # second pass: nothing rebuilt
        assert torch.equal(y0, y1)
        w = next(p for p in m.parameters() if p.dim() == 5 and p.shape[1] >= 16)
        w.mul_(1.5)                                         # in place: the version counter moves
        y2 = m(x.clone())
        assert be.form_rebuilds > n1
        be.WEIGHT_FORM_CACHE = False
        try:
            want = m(x.clone())
        finally:
            be.WEIGHT_FORM_CACHE = True
        assert torch.equal(y2, want) and not torch.equal(y2, y0)
        # a write through .data is invisible to the version counter: the documented way out is the invalidation
        w.data.mul_(2.0)
        be.invalidate_weight_forms()
        y3 = m(x.clone())
        be.WEIGHT_FORM_CACHE = False
        try:
            want3 = m(x.clone())
        finally:
            be.WEIGHT_FORM_CACHE = True
        assert torch.equal(y3, want3)


def test_forms_die_with_their_parameters():
    dev = torch.device("cuda:0")
    be.invalidate_weight_forms()
    m = _flavr(dev).eval()
    x = torch.rand(1, 2, 4, 32, 32).to(dev)
    with torch.no_grad():
        m(x.clone())
    assert len(be._forms) > 10
    torch.cuda.synchronize()
    del m
    gc.collect()
    assert len(be._forms) == 0 and len(be._form_of_panel) == 0
