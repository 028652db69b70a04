"""Fixtures for the training-patch feed, produced by the REFERENCE's own data set classes
(build container only; see tools/gen_golden.py for the import recipe).

    python tools/gen_golden_feed.py      # rewrites tests/golden/feed_*.npz, feed_misc.json

What runs is the reference's code: TrainSetMultipleSegSR.__getitem__ (utils/train_set.py:205-223),
TrainSetMultiple.__getitem__ (:330-434), TrainSetMultipleSegSREfficient.__getitem__ (:100-160), utils/pad.py,
calc_extended_patch_size (utils/blur_kernel_ops.py:21-36), zscore_normalization (utils/seg_utils.py:137-156).  The
objects are made with object.__new__ and given the attributes their __init__ would have set from the (absent-reader)
files.  Two names the reference takes from absent packages are bound to stand-ins before the calls:
  * `resize` (resize.pytorch): the tap tables of rehrseg_amd.utils.train_set.resize_taps applied with numpy -- so the
    fixtures pin everything AROUND the resize (and its call arguments), not the resize itself;
  * `train_transform` (batchgenerators chain): identity.
The volumes are regenerated in the tests from the seeds stored here; only outputs are written.
"""
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from feed_cases import EFF_CASES, KERNEL, MULTI_CASES, SEGSR_CASES, volumes_multi, volumes_seg  # noqa: E402
from gen_golden import OUT, _Finder  # noqa: E402
from rehrseg_amd.utils.train_set import resize_taps  # noqa: E402  (host-side numpy table, no GPU)


def resize_standin(x, dxyz, order=3):
    """resize(x, (dx, 1), order) on a (batch, channel, n, m) tensor, from the product's tap table (see the docstring)."""
    assert dxyz[1] == 1 and x.ndim == 4
    idx, w = resize_taps(x.shape[2], dxyz[0], order)
    a = x.numpy().astype(np.float32)
    out = np.zeros(a.shape[:2] + (idx.shape[0],) + a.shape[3:], np.float32)
    for j in range(idx.shape[0]):
        acc = np.zeros_like(out[:, :, 0])
        for t in range(idx.shape[1]):
            acc = acc + np.float32(w[j, t]) * a[:, :, idx[j, t]]
        out[:, :, j] = acc
    return torch.from_numpy(out)


def blur_numpy(vol, kernel, axis):
    """F.conv2d(..., padding='same') with an (L, 1) kernel = zero-padded cross-correlation along `axis` (eager torch, as
    the reference's load_img does at :306-318)."""
    import torch.nn.functional as F
    k = torch.from_numpy(kernel.astype(np.float32))[None, None, :, None]
    image = torch.from_numpy(vol)  # (x, y, z, 2)
    if axis == 0:
        t = image.permute(2, 3, 0, 1)[:, 0:1]
    else:
        t = image.permute(2, 3, 1, 0)[:, 0:1]
    return F.conv2d(t.contiguous(), k, padding="same").numpy()


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference")
    import utils.train_set as ts
    import utils.blur_kernel_ops as bko
    ts.resize = resize_standin

    misc = {}
    for name, (shapes, ps, sep, blur, flip, seed, draws) in MULTI_CASES.items():
        vols = volumes_multi(seed, shapes)
        ds = object.__new__(ts.TrainSetMultiple)
        ds.patch_size, ds.random_flip, ds.device, ds.preload, ds.blur = ps, flip, "cpu", True, blur
        ds.slice_separation, ds.train_transform, ds.all_subjects = float(sep), None, list(range(len(vols)))
        ds.imgs_hr = [v[..., :1] for v in vols]
        ds.labels_hr = [v[..., 1:].astype("uint8") for v in vols]
        ds.imgs_filtered_x = [blur_numpy(v, KERNEL, 0) if blur else [None] for v in vols]
        ds.imgs_filtered_y = [blur_numpy(v, KERNEL, 1) if blur else [None] for v in vols]
        random.seed(seed)
        rec, zeroed = {}, 0
        for k in range(draws):
            lr, hr = ds[k % len(vols)]
            rec[f"lr{k}"], rec[f"hr{k}"] = lr.numpy(), hr.numpy()
            zeroed += int((lr.numpy()[:, 0] == 0).all() or (lr.numpy()[:, -1] == 0).all())
        if "3d_blur" in name:
            assert zeroed > 0, "pick a seed that exercises the blank-slice branch"
        rec["filtered_x0"], rec["filtered_y0"] = (ds.imgs_filtered_x[0], ds.imgs_filtered_y[0]) if blur else (0, 0)
        np.savez_compressed(os.path.join(OUT, f"feed_{name}.npz"), **rec)
        print(name, {k: v.shape for k, v in list(rec.items())[:2]}, "blanked", zeroed)

    for name, (shapes, ps, flip, seed, draws) in SEGSR_CASES.items():
        vols = volumes_multi(seed, shapes)
        ds = object.__new__(ts.TrainSetMultipleSegSR)
        ds.patch_size, ds.random_flip, ds.split_subjects = ps, flip, list(range(len(vols)))
        ds.imgs, ds.labels = [], []
        for image in vols:  # the body of __init__ after parse_image (:183-193), through the reference's target_pad
            target = [max(s, p) for s, p in zip(image.shape[:3], ps)] + [image.shape[3], 2]
            image, _ = ts.target_pad(image, target, mode="constant")
            ds.imgs.append(image[..., :1])
            ds.labels.append(image[..., 1:].astype("uint8"))
        random.seed(seed)
        rec = {}
        for k in range(draws):
            img, lab = ds[k % len(vols)]
            rec[f"img{k}"], rec[f"lab{k}"] = img.numpy(), lab.numpy()
        np.savez_compressed(os.path.join(OUT, f"feed_{name}.npz"), **rec)
        print(name, rec["img0"].shape)

    for name, (shapes, ps, sep, unc, flip, norm, seed, draws) in EFF_CASES.items():
        vols = volumes_seg(seed, shapes)
        ds = object.__new__(ts.TrainSetMultipleSegSREfficient)
        ds.patch_size, ds.separation, ds.random_flip, ds.uncertainty, ds.norm = ps, sep, flip, unc, norm
        ds.imgs = [v["img"].copy() for v in vols]
        ds.labels = [v["seg"] for v in vols]
        ds.uncertainties = [v["uncertainty"] for v in vols]
        ds.train_transform = lambda **kw: kw
        random.seed(seed)
        rec = {}
        for k in range(draws):
            img, lab_lr, lab, u = ds[k % len(vols)]
            rec[f"img{k}"], rec[f"lab_lr{k}"], rec[f"lab{k}"] = np.asarray(img), np.asarray(lab_lr), np.asarray(lab)
            rec[f"unc{k}"] = np.asarray(u, dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, f"feed_{name}.npz"), **rec)
        print(name, rec["img0"].shape, rec["lab0"].shape)

    # utils/pad.py and calc_extended_patch_size on a few arguments
    misc["get_pads"] = [[t, d, list(ts.target_pad.__globals__["get_pads"](t, d))] for t, d in
                        ((5, 5), (8, 5), (9, 5), (4, 7), (16, 1))]
    ext = []
    for L, ps in ((5, (32, 32, 1)), (8, (16, 1, 24)), (1, (8, 8, 8))):
        e, c = bko.calc_extended_patch_size(np.zeros(L), ps)
        ext.append([L, list(ps), list(e), [[s.start, s.stop] for s in c]])
    misc["calc_extended_patch_size"] = ext
    with open(os.path.join(OUT, "feed_misc.json"), "w") as f:
        json.dump(misc, f, indent=1)


if __name__ == "__main__":
    main()
