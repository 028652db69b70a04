"""Kept weight forms (rehrseg_amd.hip_backend, "weight forms"): the packed panels and Winograd-domain weights of
nn.Parameters live across launches and are rebuilt when the parameter's version counter moves.  Keeping them must
change nothing: training steps with the cache (and its rebuild stream) equal steps without it bit for bit, frozen
networks stop rebuilding, in-place updates are seen, dead parameters release their forms."""
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


def _steps(make, x, cache, stream, n=3):
    """n fused-SGD steps (fused optimizers do not move the parameters' version counters: the optimizer-step hook of
    hip_backend is what tells the kept forms; SGD rather than Adam because Adam's 1 / sqrt(v) turns the last-bit
    differences of the atomically accumulated statistics into lr-sized ones); returns losses, parameters, rebuilds."""
    old = be.WEIGHT_FORM_CACHE, be.WEIGHT_FORM_STREAM
    be.invalidate_weight_forms()
    be.WEIGHT_FORM_CACHE, be.WEIGHT_FORM_STREAM = cache, stream
    try:
        m = make()
        opt = torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, fused=True)
        losses, rebuilds = [], []
        for _ in range(n):
            r0 = be.form_rebuilds
            opt.zero_grad()
            y = m(x.clone())
            y = y[0] if isinstance(y, (tuple, list)) else y
            loss = y.abs().mean()
            loss.backward()
            opt.step()
            losses.append(float(loss))
            rebuilds.append(be.form_rebuilds - r0)
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in m.parameters()], rebuilds
    finally:
        be.WEIGHT_FORM_CACHE, be.WEIGHT_FORM_STREAM = old
        be.invalidate_weight_forms()


@pytest.mark.parametrize("which", ["flavr", "seg"])
def test_training_steps_do_not_depend_on_kept_forms(which):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    if which == "flavr":
        x = torch.rand(2, 2, 4, 32, 32, generator=g).to(dev)
        make = lambda: _flavr(dev)
    else:
        x = torch.rand(1, 1, 32, 32, 32, generator=g).to(dev)

        def make():
            import torch.nn as nn
            torch.manual_seed(3)      # four stages with a two-source decoder: the halves of the virtual concat are forms too
            return SegModel(input_channels=1, num_classes=2, n_stages=4, upscale=4, features_per_stage=[32, 64, 128, 256],
                            conv_op=nn.Conv3d, kernel_sizes=[[3, 3, 3]] * 4, strides=[[1, 1, 1]] + [[2, 2, 2]] * 3,
                            n_conv_per_stage=[2] * 4, n_conv_per_stage_decoder=[2] * 3, conv_bias=True,
                            norm_op=nn.InstanceNorm3d, norm_op_kwargs={"eps": 1e-5, "affine": True}, dropout_op=None,
                            dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs={"inplace": True},
                            deep_supervision=False).to(dev)
    base = _steps(make, x, cache=False, stream=False)
    again = _steps(make, x, cache=False, stream=False)
    kept = _steps(make, x, cache=True, stream=False)
    side = _steps(make, x, cache=True, stream=True)
    assert base[2] == [0, 0, 0]
    assert kept[2][0] > 10 and kept[2][1] == kept[2][0] == kept[2][2]      # every form once per step, no more
    assert side[2] == kept[2]
    # The statistics epilogues accumulate with atomics: two runs WITHOUT kept forms differ in the last bits of a
    # gradient.  The bar for the runs with kept forms is that run-to-run spread (measured here) plus 2e-5 relative.
    for other in (kept, side):
        for a, b in zip(base[0], other[0]):
            assert abs(a - b) <= 1e-6 * abs(a)
        for pa, pb, pc in zip(base[1], other[1], again[1]):
            spread = float((pa - pc).abs().max())
            # (floor: conv biases in front of an InstanceNorm have a mathematically zero gradient -- their values are
            # rounding residue of ~1e-5 and move with the order of the atomics, which depends on launch timing)
            assert float((pa - pb).abs().max()) <= 4 * spread + 2e-5 * max(float(pa.abs().max()), 1e-2)


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
