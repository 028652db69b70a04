"""Whole-volume inference paths on the MI355X: the same reference fixtures as tests/test_inference_cpu.py
through the HIP kernels, plus size-independent properties at a larger volume."""
import pytest
import torch

from rehrseg_amd.utils import seg_utils as su
from rehrseg_amd.utils import sr_utils as sr
from test_inference_cpu import _flavr, check_vol

pytestmark = pytest.mark.gpu


def test_apply_to_vol_flavr_gpu_matches_reference():
    check_vol(_flavr("cuda:0"), "cuda:0", 1e-4)


def test_apply_to_vol_flavr_batching_is_invisible():
    """A 40-slice 64x48 volume: one window per call == 32 windows per call, and the first / last windows see
    the zero slices the reference pads with."""
    dev = torch.device("cuda:0")
    m = _flavr(dev)
    g = torch.Generator().manual_seed(3)
    vol = torch.rand(40, 2, 64, 48, generator=g).to(dev)
    a = sr.apply_to_vol_flavr(m, vol.clone(), 0, window_batch=1)
    b = sr.apply_to_vol_flavr(m, vol.clone(), 0, window_batch=32)
    assert tuple(a.shape) == (4 * 39, 2, 48, 64)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


def test_mirror_tta_batched_equals_eight_calls_on_segmodel():
    from test_segmodel_cpu import SMALL, build   # the small SegModel plan of the parity tests
    dev = torch.device("cuda:0")
    m = build(SMALL, dev)[0].eval()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 1, 16, 32, 32, generator=g).to(dev)
    with torch.no_grad():
        got = su._internal_maybe_mirror_and_predict(m, x.clone(), 1, deep_supervision=False)
        import itertools
        combos = [c for i in range(3) for c in itertools.combinations([2, 3, 4], i + 1)]
        ref = m(x.clone())[1]
        for axes in combos:
            ref = ref + torch.flip(m(torch.flip(x, axes))[1], axes)
        ref = ref / 8
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_tiled_predictor_reference_fixture_on_device():
    """The reference's own tiled-predictor fixture (tests/golden/inference_paths.npz, produced by its
    _internal_predict_sliding_window_return_logits over the toy net) with the data, the fp16 accumulators and the
    blend on the device (ref utils/seg_utils.py:240-287)."""
    import numpy as np
    from test_inference_cpu import G
    from toy_models import ToySegNet
    dev = torch.device("cuda:0")
    net = ToySegNet(sep=2).to(dev)
    data = torch.from_numpy(G["tile_in"]).to(dev)
    patch = [6, 12, 10]
    sl = su._internal_get_sliding_window_slicers(data.shape[1:], patch_size=patch)
    lr = su._internal_predict_sliding_window_return_logits(data.clone(), sl, net, True, 0, 1, patch,
                                                           use_gaussian=False, deep_supervision=False)
    hr = su._internal_predict_sliding_window_return_logits(data.clone(), sl, net, True, 1, 2,
                                                           [patch[0] * 2, patch[1], patch[2]])
    assert lr.is_cuda and hr.is_cuda and lr.dtype == torch.half
    assert np.allclose(lr.float().cpu().numpy(), G["tile_lr"], rtol=2e-3, atol=2e-3)
    assert np.allclose(hr.float().cpu().numpy(), G["tile_hr"], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("out_idx", [0, 1])
def test_tiled_predictor_gaussian_blend_with_hip_segmodel(out_idx):
    """_internal_predict_sliding_window_return_logits + Gaussian blending + 8x mirror TTA with the
    HIP SegModel on the device (ref utils/seg_utils.py:201-287), against the same procedure composed independently on
    the CPU: the oracle SegModel evaluated on every tile and every mirroring one call at a time (the reference's loop),
    an importance map built here from scipy (nnunetv2's compute_gaussian recipe: unpinned third party) and the
    reference's fp16 accumulators.  Tolerance 2e-3 of the logits' scale (one fp16 rounding of the accumulators)."""
    import itertools
    import numpy as np
    from scipy.ndimage import gaussian_filter
    from oracle import segmodel_oracle as so
    from oracle.detinit import det_input
    from test_segmodel_cpu import SMALL, build
    dev = torch.device("cuda:0")
    m, sd = build(SMALL, dev)
    m.eval()
    sep = SMALL["upscale"] if out_idx == 1 else 1
    data = det_input("tiled.vol", (1, 12, 44, 40), "randn")
    patch = [8, 32, 32]
    sl = su._internal_get_sliding_window_slicers(data.shape[1:], patch_size=patch)
    assert len(sl) >= 8                                              # several overlapping tiles along every axis
    psz = [patch[0] * sep, patch[1], patch[2]]
    with torch.no_grad():
        got = su._internal_predict_sliding_window_return_logits(data.clone().to(dev), sl, m, True, out_idx, sep, psz,
                                                                use_gaussian=True, deep_supervision=False)
    assert got.is_cuda and got.dtype == torch.half
    # independent composition, in the reference's arithmetic (:240-287): importance map and both accumulators in fp16
    tmp = np.zeros(psz)
    tmp[tuple(i // 2 for i in psz)] = 1
    g = torch.from_numpy(gaussian_filter(tmp, [i / 8.0 for i in psz], 0, mode="constant", cval=0))
    g = (g / (g.max() / 10)).to(torch.float16)
    g[g == 0] = g[g != 0].min()
    osd = {k: v for k, v in sd.items() if k in so.segmodel_shapes(SMALL)}
    acc = torch.zeros(2, data.shape[1] * sep, data.shape[2], data.shape[3], dtype=torch.float16)
    cnt = torch.zeros(data.shape[1] * sep, data.shape[2], data.shape[3], dtype=torch.float16)
    combos = [c for i in range(3) for c in itertools.combinations([2, 3, 4], i + 1)]
    with torch.no_grad():
        for s in sl:
            x = data[s][None]
            pred = so.seg_model(osd, x, SMALL)[out_idx]
            for axes in combos:
                pred = pred + torch.flip(so.seg_model(osd, torch.flip(x, axes), SMALL)[out_idx], axes)
            pred = pred[0] / 8
            msl = (slice(None), slice(s[1].start * sep, s[1].stop * sep), s[2], s[3])
            acc[msl] += pred * g
            cnt[msl[1:]] += g
    ref = (acc / cnt).float()
    # Voxels that only ever see the far corner of a tile carry weights of a few fp16 subnormal units (the map spans
    # 10 .. 6e-8): there `prediction * gaussian` has no significant bits left -- in the reference's arithmetic as much
    # as here -- and a 1e-6 difference in the prediction flips the quotient.  They are compared separately.
    solid = cnt.float() >= 1e-3
    d = (got.float().cpu() - ref).abs()
    err = float(d[:, solid].max() / ref[:, solid].abs().max())
    print(f"tiled predictor (out_idx {out_idx}, {len(sl)} tiles): max-rel vs the CPU composition {err:.2e} on "
          f"{float(solid.float().mean()):.3f} of the voxels; {int((~solid).sum())} voxels with total weight < 1e-3")
    assert float(solid.float().mean()) >= 0.75
    assert err <= 2e-3
    lab_agree = float((got.float().cpu().argmax(0) == ref.argmax(0))[solid].float().mean())
    assert lab_agree >= 0.999, lab_agree
    assert torch.isfinite(got.float()).all()
