"""Whole-volume inference helpers against fixtures captured from the reference's own
functions (tools/gen_golden_inference.py): window construction of apply_to_vol_flavr (all windows batched
here, one call per window there), sliding-window geometry, mirror TTA (8 variants as one batch here) and the
fp16-accumulating tiled predictor.  The network under apply_to_vol_flavr is the product UNet_3D_3D with the
C-ABI emulated on the CPU."""
import os

import numpy as np
import pytest
import torch

from oracle.detinit import det_tensor
from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
from rehrseg_amd.utils import seg_utils as su
from rehrseg_amd.utils import sr_utils as sr
from toy_models import ToySegNet

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "inference_paths.npz"))


def _flavr(device="cpu"):
    m = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True).eval()
    m.load_state_dict({k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()})
    return m.to(device)


def check_vol(model, device, tol):
    vol = torch.from_numpy(G["vol_in"]).to(device)
    for idx, key in ((0, "vol_out0"), (1, "vol_out1")):
        for wb in (2, 32):   # several batches / one batch
            out = sr.apply_to_vol_flavr(model, vol.clone(), idx, window_batch=wb)
            assert out.device.type == "cpu" and tuple(out.shape) == G[key].shape
            scale = float(np.abs(G[key]).max())
            assert float(np.abs(out.numpy() - G[key]).max()) <= tol * scale


def test_apply_to_vol_flavr_matches_reference(emu):
    check_vol(_flavr(), "cpu", 1e-4)


def test_window_indices_edge_cases():
    assert sr._window_indices(2) == [[-1, -1, 0, 1]]                       # ref: front-pad the short first window
    assert sr._window_indices(3) == [[-1, 0, 1, 2], [0, 1, 2, -1]]
    assert sr._window_indices(5)[1:3] == [[0, 1, 2, 3], [1, 2, 3, 4]]


def test_sliding_window_geometry():
    cases = [((20, 45, 63), (14, 32, 48), 0.5), ((14, 320, 384), (14, 320, 384), 0.5), ((9, 70, 33), (8, 32, 32), 0.75)]
    for i, (img, tile, step) in enumerate(cases):
        steps = su.compute_steps_for_sliding_window(img, tile, step)
        for a in range(3):
            assert steps[a] == G[f"steps{i}_{a}"].tolist()
        sl = su._internal_get_sliding_window_slicers(img, patch_size=list(tile), tile_step_size=step)
        got = [[s.start for s in t[1:]] + [s.stop for s in t[1:]] for t in sl]
        assert got == G[f"slicers{i}"].tolist()


def test_mirror_tta_batched_equals_reference_loop():
    net = ToySegNet(sep=2)
    x = torch.from_numpy(G["tta_in"])
    lr = su._internal_maybe_mirror_and_predict(net, x.clone(), 0, deep_supervision=False)
    hr = su._internal_maybe_mirror_and_predict(net, x.clone(), 1, deep_supervision=False)
    assert np.allclose(lr.numpy(), G["tta_lr"], rtol=1e-6, atol=1e-6)
    assert np.allclose(hr.numpy(), G["tta_hr"], rtol=1e-6, atol=1e-6)


def test_tiled_predictor_matches_reference():
    net = ToySegNet(sep=2)
    data = torch.from_numpy(G["tile_in"])
    patch = [6, 12, 10]
    sl = su._internal_get_sliding_window_slicers(data.shape[1:], patch_size=patch)
    lr = su._internal_predict_sliding_window_return_logits(data.clone(), sl, net, False, 0, 1, patch,
                                                           use_gaussian=False, deep_supervision=False)
    hr = su._internal_predict_sliding_window_return_logits(data.clone(), sl, net, False, 1, 2,
                                                           [patch[0] * 2, patch[1], patch[2]])
    assert lr.dtype == torch.half and hr.dtype == torch.half
    # fp16 accumulators: one half ulp at the logits' magnitude
    assert np.allclose(lr.float().numpy(), G["tile_lr"], rtol=2e-3, atol=2e-3)
    assert np.allclose(hr.float().numpy(), G["tile_hr"], rtol=2e-3, atol=2e-3)


def test_gaussian_importance_map_properties():
    """compute_gaussian is a restatement of an absent nnunetv2 function (unpinned): check its defining
    properties only -- peak = value_scaling_factor at the centre, symmetric, strictly positive."""
    g = su.compute_gaussian((6, 12, 10), sigma_scale=1. / 8, value_scaling_factor=10, dtype=torch.float32)
    assert abs(float(g.max()) - 10.0) < 1e-5 and float(g.min()) > 0
    assert float(g[3, 6, 5]) == float(g.max())
    assert torch.allclose(g[1:], torch.flip(g[1:], (0,)), atol=1e-6)
    su.compute_gaussian.cache_clear()
