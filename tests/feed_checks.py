"""The fixture comparisons of the patch feed, shared by the CPU run (kernel semantics emulated in numpy, checks the
host-side descriptors) and the GPU run (the HIP kernels through the C-ABI)."""
import json
import os
import random

import numpy as np
import torch

from feed_cases import EFF_CASES, KERNEL, MULTI_CASES, SEGSR_CASES, volumes_multi, volumes_seg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def check_multi(name, device, batched=False):
    from rehrseg_amd.utils.train_set import TrainSetMultiple
    shapes, ps, sep, blur, flip, seed, draws = MULTI_CASES[name]
    g = np.load(os.path.join(GOLD, f"feed_{name}.npz"))
    vols = volumes_multi(seed, shapes)
    ds = TrainSetMultiple(None, list(range(len(vols))), sep, 1.0, None, None, ps, flip, device, blur=blur, volumes=vols,
                          blur_kernel=KERNEL)
    if blur:  # the slice-profile blur of load_img (:306-318), built on the device from the raw volume
        np.testing.assert_allclose(_np(ds.imgs_filtered_x[0]), g["filtered_x0"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(_np(ds.imgs_filtered_y[0]), g["filtered_y0"], rtol=1e-6, atol=1e-6)
    random.seed(seed)
    if batched:  # the same draws, one launch per stage for each group of equally shaped items
        outs = []
        for k in range(0, draws, 2):
            lr, hr = ds.batch([k % len(vols), (k + 1) % len(vols)])
            outs += [(lr[0], hr[0]), (lr[1], hr[1])]
    else:
        outs = [ds[k % len(vols)] for k in range(draws)]
    for k, (lr, hr) in enumerate(outs):
        assert tuple(lr.shape) == g[f"lr{k}"].shape and tuple(hr.shape) == g[f"hr{k}"].shape, (k, lr.shape, hr.shape)
        np.testing.assert_array_equal(_np(hr), g[f"hr{k}"], err_msg=f"{name} hr{k}")
        # the label channel and the blank slices are exact; the image channel went through the cubic taps
        np.testing.assert_array_equal(_np(lr)[1], g[f"lr{k}"][1], err_msg=f"{name} lr{k} label")
        np.testing.assert_allclose(_np(lr)[0], g[f"lr{k}"][0], rtol=1e-6, atol=1e-6, err_msg=f"{name} lr{k}")


def check_segsr(name, device):
    from rehrseg_amd.utils.train_set import TrainSetMultipleSegSR
    shapes, ps, flip, seed, draws = SEGSR_CASES[name]
    g = np.load(os.path.join(GOLD, f"feed_{name}.npz"))
    vols = volumes_multi(seed, shapes)
    ds = TrainSetMultipleSegSR(None, list(range(len(vols))), 4.0, 1.0, ps, flip, device=device, volumes=vols)
    random.seed(seed)
    for k in range(draws):
        img, lab = ds[k % len(vols)]
        np.testing.assert_array_equal(_np(img), g[f"img{k}"], err_msg=f"{name} img{k}")
        np.testing.assert_array_equal(_np(lab), g[f"lab{k}"], err_msg=f"{name} lab{k}")


def check_eff(name, device, batched=False):
    from rehrseg_amd.utils.train_set import TrainSetMultipleSegSREfficient
    shapes, ps, sep, unc, flip, norm, seed, draws = EFF_CASES[name]
    g = np.load(os.path.join(GOLD, f"feed_{name}.npz"))
    vols = volumes_seg(seed, shapes)
    ds = TrainSetMultipleSegSREfficient(None, list(range(len(vols))), float(sep), 1.0, ps, None, flip, unc, norm=norm,
                                        device=device, volumes=vols)
    random.seed(seed)
    if batched:
        outs = []
        for k in range(0, draws, 2):
            b = ds.batch([k % len(vols), (k + 1) % len(vols)])
            outs += [tuple(o[j] if torch.is_tensor(o) else o for o in b) for j in range(2)]
    else:
        outs = [ds[k % len(vols)] for k in range(draws)]
    for k, (img, lab_lr, lab, u) in enumerate(outs):
        # the reference re-normalises its (already normalised) stored volume in place on every access (:104-105):
        # a few ulp of drift per access, hence not bit-equal
        np.testing.assert_allclose(_np(img), g[f"img{k}"], rtol=1e-5, atol=1e-5, err_msg=f"{name} img{k}")
        np.testing.assert_array_equal(_np(lab_lr), g[f"lab_lr{k}"], err_msg=f"{name} lab_lr{k}")
        np.testing.assert_array_equal(_np(lab), g[f"lab{k}"], err_msg=f"{name} lab{k}")
        np.testing.assert_allclose(_np(u), g[f"unc{k}"], rtol=1e-6, atol=1e-6, err_msg=f"{name} unc{k}")


def check_misc():
    from rehrseg_amd.utils.blur_kernel_ops import calc_extended_patch_size
    from rehrseg_amd.utils.pad import get_pads
    with open(os.path.join(GOLD, "feed_misc.json")) as f:
        m = json.load(f)
    for t, d, want in m["get_pads"]:
        assert list(get_pads(t, d)) == want
    for L, ps, e, c in m["calc_extended_patch_size"]:
        e2, c2 = calc_extended_patch_size(np.zeros(L), tuple(ps))
        assert list(e2) == e and [[s.start, s.stop] for s in c2] == c


def check_loader(device):
    """the three data sets through `torch.utils.data.DataLoader(num_workers=0, pin_memory=False)` with the
    default collate (how a reference-style loop consumes them, train_all.py:294-301,502-509 minus the workers):
    batched shapes and values against the fixtures of the same draws."""
    from torch.utils.data import DataLoader
    from rehrseg_amd.utils.train_set import TrainSetMultiple, TrainSetMultipleSegSR, TrainSetMultipleSegSREfficient
    # stage 1
    name = "multi_2d_blur"
    shapes, ps, sep, blur, flip, seed, draws = MULTI_CASES[name]
    g = np.load(os.path.join(GOLD, f"feed_{name}.npz"))
    vols = volumes_multi(seed, shapes)
    ds = TrainSetMultiple(None, list(range(len(vols))), sep, 1.0, None, None, ps, flip, device, blur=blur, volumes=vols,
                          blur_kernel=KERNEL)
    random.seed(seed)
    (lr, hr), = list(DataLoader(ds, batch_size=len(vols), shuffle=False, num_workers=0, pin_memory=False))
    assert lr.device.type == torch.device(device).type
    assert tuple(lr.shape) == (len(vols),) + g["lr0"].shape and tuple(hr.shape) == (len(vols),) + g["hr0"].shape
    for k in range(len(vols)):
        np.testing.assert_array_equal(_np(hr[k]), g[f"hr{k}"])
        np.testing.assert_allclose(_np(lr[k]), g[f"lr{k}"], rtol=1e-6, atol=1e-6)
    # stage 2, efficient set (image, LR label, HR label, uncertainty)
    name = "eff_unc"
    shapes, ps, sep, unc, flip, norm, seed, draws = EFF_CASES[name]
    g = np.load(os.path.join(GOLD, f"feed_{name}.npz"))
    vols = volumes_seg(seed, shapes)
    ds = TrainSetMultipleSegSREfficient(None, list(range(len(vols))), float(sep), 1.0, ps, None, flip, unc, norm=norm,
                                        device=device, volumes=vols)
    random.seed(seed)
    (img, lab_lr, lab, u), = list(DataLoader(ds, batch_size=len(vols), shuffle=False, num_workers=0, pin_memory=False))
    assert tuple(img.shape) == (len(vols),) + g["img0"].shape and tuple(lab.shape) == (len(vols),) + g["lab0"].shape
    for k in range(len(vols)):
        np.testing.assert_allclose(_np(img[k]), g[f"img{k}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(_np(lab_lr[k]), g[f"lab_lr{k}"])
        np.testing.assert_array_equal(_np(lab[k]), g[f"lab{k}"])
        np.testing.assert_allclose(_np(u[k]), g[f"unc{k}"], rtol=1e-6, atol=1e-6)
    # stage 2, plain set: items of one subject at a time (the subjects' patch shapes differ)
    name = "segsr"
    shapes, ps, flip, seed, draws = SEGSR_CASES[name]
    g = np.load(os.path.join(GOLD, f"feed_{name}.npz"))
    vols = volumes_multi(seed, shapes)
    ds = TrainSetMultipleSegSR(None, list(range(len(vols))), 4.0, 1.0, ps, flip, device=device, volumes=vols)
    random.seed(seed)
    for k, (img, lab) in enumerate(DataLoader(ds, batch_size=1, shuffle=False, num_workers=0, pin_memory=False)):
        np.testing.assert_array_equal(_np(img[0]), g[f"img{k}"])
        np.testing.assert_array_equal(_np(lab[0]), g[f"lab{k}"])
    assert k == len(vols) - 1


def check_worker_guard(device, monkeypatch):
    """A DataLoader worker process must get a clear error instead of touching the GPU runtime it inherited."""
    import torch.utils.data
    from rehrseg_amd import lib
    from rehrseg_amd.utils.train_set import TrainSetMultipleSegSR
    shapes, ps, flip, seed, draws = SEGSR_CASES["segsr"]
    vols = volumes_multi(seed, shapes)
    ds = TrainSetMultipleSegSR(None, list(range(len(vols))), 4.0, 1.0, ps, flip, device=device, volumes=vols)
    monkeypatch.setattr(torch.utils.data, "get_worker_info", lambda: object())
    try:
        ds[0]
    except lib.RehrsegHipError as e:
        assert "num_workers=0" in str(e)
    else:
        raise AssertionError("a worker-side __getitem__ must raise")
