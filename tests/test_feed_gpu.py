"""The patch-feed kernels (rehr_patch_gather, rehr_axis_resample_f32) through the C-ABI on the fixtures generated from
the reference's data set classes, plus size-independent checks at training size."""
import random

import numpy as np
import pytest
import torch

import feed_checks
from feed_cases import BATCHABLE, EFF_CASES, MULTI_CASES, SEGSR_CASES, emu_axis_resample, emu_patch_gather

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", sorted(MULTI_CASES))
def test_train_set_multiple(name):
    feed_checks.check_multi(name, DEV)


@pytest.mark.parametrize("name", BATCHABLE)
def test_train_set_multiple_batched(name):
    feed_checks.check_multi(name, DEV, batched=True)


@pytest.mark.parametrize("name", sorted(SEGSR_CASES))
def test_train_set_segsr(name):
    feed_checks.check_segsr(name, DEV)


@pytest.mark.parametrize("name", sorted(EFF_CASES))
@pytest.mark.parametrize("batched", [False, True])
def test_train_set_efficient(name, batched):
    feed_checks.check_eff(name, DEV, batched)


@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8])
def test_gather_kernel_vs_semantics(dtype):
    """Random descriptors (every axis order, negative strides, pad boxes, more items than one launch holds, both the
    plain and the LDS-transposing kernel) against the numpy statement of the C-ABI semantics."""
    from rehrseg_amd import hip_backend as hb
    from rehrseg_amd.utils.train_set import View
    rng = np.random.RandomState(3)
    for trial in range(12):
        shape = tuple(int(v) for v in rng.randint(3, 70, size=3)) + (int(rng.randint(1, 4)),)
        vol = rng.randint(0, 255, size=shape).astype(np.float32)
        src = torch.from_numpy(vol).to(dtype).to(DEV)
        ps = [int(rng.randint(1, s + 6)) for s in shape[:3]]
        views = []
        for _ in range(int(rng.randint(1, 40))):
            v = View(shape)
            st = [int(rng.randint(0, max(s - p, 0) + 1)) for s, p in zip(shape[:3], ps)]
            v = v.crop(st + [0], ps + [shape[3]]).target_pad(ps + [shape[3]])
            for ax in range(3):
                if rng.rand() < 0.5:
                    v = v.flip(ax)
            views.append(v.transpose(tuple(int(k) for k in rng.permutation(4))) if trial % 2 else v.transpose(3, 2, 1, 0))
        shapes = {tuple(v.shape) for v in views}
        views = [v for v in views if tuple(v.shape) == tuple(views[0].shape)]
        items = [v.item(src)[0] for v in views]
        dims = views[0].item(src)[1]
        got = hb.patch_gather(items, dims, 0.5, 2.0).cpu()
        want = emu_patch_gather([(src.cpu(),) + it[1:] for it in items], dims, 0.5, 2.0)
        assert torch.equal(got, want), (trial, shape, ps, len(shapes))


def test_axis_resample_kernel_vs_semantics():
    from rehrseg_amd import hip_backend as hb
    from rehrseg_amd.utils.train_set import blur_taps, resize_taps
    rng = np.random.RandomState(4)
    x = torch.from_numpy(rng.rand(3, 37, 5, 11).astype(np.float32)).to(DEV)
    for axis in range(4):
        n = x.shape[axis]
        for table in (resize_taps(n, 4.0, 3), resize_taps(n, 2.5, 1), resize_taps(n, 3.0, 0), blur_taps(n, rng.rand(7))):
            idx, w = (torch.from_numpy(t) for t in table)
            got = hb.axis_resample(x, axis, idx.to(DEV), w.to(DEV)).cpu()
            want = emu_axis_resample(x.cpu(), axis, idx, w)
            torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)


def test_gather_rejects_out_of_volume():
    from rehrseg_amd import hip_backend as hb, lib
    src = torch.zeros(4, 4, 4, device=DEV)
    with pytest.raises(lib.RehrsegHipError):
        hb.patch_gather([(src, 0, [16, 4, 1, 0], [0] * 4, [5, 4, 4, 1])], (5, 4, 4, 1))
    with pytest.raises(lib.RehrsegHipError):
        hb.patch_gather([(src.cpu(), 0, [16, 4, 1, 0], [0] * 4, [4, 4, 4, 1])], (4, 4, 4, 1))


def test_training_size_properties():
    """Stage-2 patches at training size: every draw is a crop of the volume (checked by locating it), the LR view is
    every k-th slice of the HR view, flips are involutions of the un-flipped draw, pads are zeros."""
    from rehrseg_amd.utils.train_set import TrainSetMultipleSegSREfficient
    rng = np.random.RandomState(5)
    shape = (300, 280, 150)
    vol = dict(img=rng.rand(*shape).astype(np.float32), seg=(rng.rand(*shape) > 0.5).astype(np.uint8),
               uncertainty=rng.randint(0, 256, size=shape).astype(np.uint8))
    ps, sep = (192, 192, 40), 4   # z: 160 > 150 -> padded
    ds = TrainSetMultipleSegSREfficient(None, [0], float(sep), 1.0, ps, None, False, True, norm=False, device=DEV,
                                        volumes=[vol])
    state = random.getstate()
    random.seed(9)
    img, lab_lr, lab, unc = ds[0]
    random.seed(9)
    x0, y0 = random.randint(0, shape[0] - ps[0]), random.randint(0, shape[1] - ps[1])
    random.setstate(state)
    assert img.shape == (1, 40, 192, 192) and lab.shape == (1, 160, 192, 192)
    want = np.zeros((ps[0], ps[1], 160), np.float32)
    want[:, :, 5:155] = vol["seg"][x0:x0 + ps[0], y0:y0 + ps[1], :]
    np.testing.assert_array_equal(lab[0].cpu().numpy(), want.transpose(2, 1, 0))
    np.testing.assert_array_equal(lab_lr[0].cpu().numpy(), want.transpose(2, 1, 0)[::sep])
    wi = np.zeros((ps[0], ps[1], 160), np.float32)
    wi[:, :, 5:155] = vol["img"][x0:x0 + ps[0], y0:y0 + ps[1], :]
    np.testing.assert_array_equal(img[0].cpu().numpy(), wi.transpose(2, 1, 0)[::sep])
    wu = np.zeros((ps[0], ps[1], 160), np.float64)
    wu[:, :, 5:155] = vol["uncertainty"][x0:x0 + ps[0], y0:y0 + ps[1], :]
    np.testing.assert_allclose(unc[0].cpu().numpy(), (1 - wu / 255.0 * 0.99).transpose(2, 1, 0)[::sep], rtol=1e-6)
    # flipped draws: same offsets (same seed), reversed axes
    ds.random_flip = True
    random.seed(9)
    img_f, _, lab_f, _ = ds[0]
    random.seed(9)
    random.randint(0, shape[0] - ps[0]), random.randint(0, shape[1] - ps[1]), random.randint(0, 0)
    fl = [random.random() < 0.5 for _ in range(3)]
    random.setstate(state)
    dims = [3 - k for k in range(3) if fl[k]]  # (x, y, z) -> axes of (1, z, y, x)
    np.testing.assert_array_equal(torch.flip(lab_f, dims).cpu().numpy() if dims else lab_f.cpu().numpy(),
                                  lab.cpu().numpy())


def test_through_a_dataloader():
    """num_workers=0 / pin_memory=False: the only loader configuration device-resident data sets admit."""
    feed_checks.check_loader(DEV)
