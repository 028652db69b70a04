"""SegModel on the MI355X kernels vs the functional oracle (bases: parity UNPINNED).
Includes BASELINE.json configs[0]'s shape (1x1x64^3, isotropic 3d_fullres plan)."""
import pytest
import torch

from oracle import segmodel_oracle as so
from test_segmodel_cpu import SMALL, build, check

pytestmark = pytest.mark.gpu


def test_segmodel_small_plan_gpu():
    check(SMALL, (2, 1, 4, 16, 16), "cuda:0", 1e-3 / 10)


def test_segmodel_aniso_plan_gpu():
    check(so.ANISO_PLAN, (1, 1, 16, 96, 96), "cuda:0", 1e-3 / 10)


def test_segmodel_cfg1_iso_plan_64cube_gpu():
    """configs[0]: 1x1x64^3 through the isotropic plan; shapes (1,2,64,64,64) and (1,2,256,64,64)."""
    m, sd = build(so.ISO_PLAN, "cuda:0")
    from oracle.detinit import det_input
    x = det_input("cfg1.x", (1, 1, 64, 64, 64), "randn")
    out, out_up = m(x.clone().cuda())
    assert tuple(out.shape) == (1, 2, 64, 64, 64) and tuple(out_up.shape) == (1, 2, 256, 64, 64)
    (out.float().mean() + out_up.float().mean()).backward()
    assert torch.isfinite(out).all() and torch.isfinite(out_up).all()
    osd = {k: v.double() for k, v in sd.items() if k in so.segmodel_shapes(so.ISO_PLAN)}
    with torch.no_grad():
        r_out, r_up = so.seg_model(osd, x.double(), so.ISO_PLAN)
    for a, b in ((out, r_out), (out_up, r_up)):
        assert float((a.detach().cpu().double() - b).abs().max() / b.abs().max()) < 1e-3
    # label maps: exact where the logit margin is clear of the numerical noise, mismatches reported otherwise
    for a, b in ((out, r_out), (out_up, r_up)):
        la, lb = a.detach().cpu().argmax(1), b.argmax(1)
        margin = (b[:, 0] - b[:, 1]).abs()
        clear = margin > 1e-3 * float(b.abs().max())
        assert bool((la == lb)[clear].all())
        print("argmax mismatches on near-ties:", int((la != lb).sum()), "of", la.numel())
