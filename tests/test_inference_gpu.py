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
