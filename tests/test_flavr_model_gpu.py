"""UNet_3D_3D on the MI355X kernels against the fixtures captured from the
reference (tolerance: 1e-3 relative, the bar BASELINE.json's north_star states)
and against the oracle at a larger shape."""
import os

import numpy as np
import pytest
import torch

from test_flavr_model_cpu import GOLD, build, check_against_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["c2_n4", "c2_n4_unc", "c1_n8"])
def test_flavr_gpu_matches_reference_fixture(tag):
    g = np.load(os.path.join(GOLD, f"flavr_{tag}.npz"))
    m, unc = build(g, "cuda:0")
    check_against_golden(g, m, unc, "cuda:0", 1e-3 / 10)  # helper allows 10x on gradients


def test_flavr_gpu_vs_oracle_batch_and_odd_extent():
    """Batch 2, reference-like 4x48x40 patch (non power-of-two extents -> masked tiles)."""
    from oracle import flavr_oracle as fo
    from oracle.detinit import det_input, det_tensor
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    m = UNet_3D_3D(2, "unet_18", 4, 4)
    sd = {k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.cuda()
    x = det_input("odd.x", (2, 2, 4, 48, 40), "rand")
    out = m(x.clone().cuda())
    loss = out.abs().mean()
    loss.backward()
    osd = {k: v.clone().requires_grad_() for k, v in sd.items()}
    ref = fo.unet_3d_3d(osd, x.clone(), 2, 4, 4)
    ref.abs().mean().backward()
    assert float((out.detach().cpu() - ref.detach()).abs().max() / ref.detach().abs().max()) < 1e-3
    for k, p in m.named_parameters():
        if osd[k].grad is None:
            continue
        n = float(osd[k].grad.norm())
        assert float((p.grad.cpu() - osd[k].grad).norm()) <= 1e-3 * n + 1e-9, k
