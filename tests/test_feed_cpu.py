"""Host side of the patch feed against fixtures generated from the reference's own data set
classes: random-number protocol, crop / pad / flip / transposition / blank-slice / permute descriptors.  The two GPU
launches are bound to a numpy emulation of their C-ABI semantics (tests/feed_cases.py) -- this checks the descriptors
the host builds; tests/test_feed_gpu.py checks the kernels on the same fixtures."""
import numpy as np
import pytest
import torch

import feed_checks
from feed_cases import BATCHABLE, EFF_CASES, MULTI_CASES, SEGSR_CASES, emu_axis_resample, emu_patch_gather


@pytest.fixture
def emulated(monkeypatch):
    from rehrseg_amd import hip_backend as hb
    from rehrseg_amd.utils import train_set as ts
    monkeypatch.setattr(hb, "patch_gather", emu_patch_gather)
    monkeypatch.setattr(hb, "axis_resample", emu_axis_resample)
    monkeypatch.setattr(ts._DeviceSet, "_check_device", lambda self, device: torch.device("cpu"))


def test_feed_refuses_cpu():
    from rehrseg_amd import lib
    from rehrseg_amd.utils.train_set import TrainSetMultipleSegSR
    with pytest.raises(lib.RehrsegHipError):
        TrainSetMultipleSegSR(None, [0], 4.0, 1.0, (4, 4, 4), device="cpu", volumes=[np.zeros((4, 4, 4, 2), np.float32)])


@pytest.mark.parametrize("name", sorted(MULTI_CASES))
def test_train_set_multiple(emulated, name):
    feed_checks.check_multi(name, "cpu")


@pytest.mark.parametrize("name", BATCHABLE)
def test_train_set_multiple_batched(emulated, name):
    feed_checks.check_multi(name, "cpu", batched=True)


def test_batch_of_unequal_shapes_is_refused(emulated):
    """(c, x, z, y) and (c, x, y, z) items with y != z cannot be stacked (the reference's collate fails there too)."""
    import random
    from feed_cases import KERNEL, volumes_multi
    from rehrseg_amd.utils.train_set import TrainSetMultiple
    vols = volumes_multi(1, [(20, 12, 9)])
    ds = TrainSetMultiple(None, [0], 4.0, 1.0, None, None, (16, 8, 8), True, "cpu", volumes=vols, blur_kernel=KERNEL)
    random.seed(0)
    with pytest.raises(ValueError):
        for _ in range(8):
            ds.batch([0, 0])


@pytest.mark.parametrize("name", sorted(SEGSR_CASES))
def test_train_set_segsr(emulated, name):
    feed_checks.check_segsr(name, "cpu")


@pytest.mark.parametrize("name", sorted(EFF_CASES))
@pytest.mark.parametrize("batched", [False, True])
def test_train_set_efficient(emulated, name, batched):
    feed_checks.check_eff(name, "cpu", batched)


def test_pad_and_extended_patch():
    feed_checks.check_misc()


def test_view_algebra_matches_numpy():
    """Random chains of the View operations against the same chain on a numpy array."""
    from rehrseg_amd.utils.train_set import View
    rng = np.random.RandomState(0)
    for trial in range(200):
        shape = tuple(int(v) for v in rng.randint(1, 7, size=3))
        a = rng.rand(*shape).astype(np.float32)
        a0 = torch.from_numpy(a.copy())
        v = View(shape)
        for _ in range(rng.randint(1, 7)):
            op = rng.randint(0, 6)
            ax = int(rng.randint(0, a.ndim))
            if op == 0:
                order = tuple(int(k) for k in rng.permutation(a.ndim))
                a, v = a.transpose(order), v.transpose(order)
            elif op == 1:
                s, e = sorted(int(k) for k in rng.randint(0, a.shape[ax] + 3, size=2))
                sl = [slice(None)] * a.ndim
                sl[ax] = slice(s, e)
                a, v = a[tuple(sl)], v.slice(ax, s, e)
            elif op == 2:
                b, c = int(rng.randint(0, 3)), int(rng.randint(0, 3))
                pads = [(0, 0)] * a.ndim
                pads[ax] = (b, c)
                a, v = np.pad(a, pads, mode="constant"), v.pad(ax, b, c)
            elif op == 3:
                a, v = np.flip(a, ax), v.flip(ax)
            elif op == 4:
                k, s = int(rng.randint(1, 4)), int(rng.randint(0, 3))
                sl = [slice(None)] * a.ndim
                sl[ax] = slice(s, None, k)
                a, v = a[tuple(sl)], v.step(ax, k, s)
            elif a.shape[ax] > 0:
                a = a.copy()
                sl = [slice(None)] * a.ndim
                which = int(rng.randint(0, 2)) - 1
                sl[ax] = slice(0, 1) if which == 0 else slice(-1, None)
                a[tuple(sl)] = 0
                v = v.zero(ax, which)
            if 0 in a.shape:
                break
        if 0 in a.shape:
            assert list(a.shape) == v.shape
            continue
        assert list(a.shape) == v.shape, trial
        item, dims = v.item(a0)
        got = emu_patch_gather([item], dims)[0].numpy().reshape(v.shape)
        np.testing.assert_array_equal(got, a, err_msg=f"trial {trial}")



def test_through_a_dataloader(emulated):
    feed_checks.check_loader("cpu")


def test_worker_processes_are_refused(emulated, monkeypatch):
    feed_checks.check_worker_guard("cpu", monkeypatch)
