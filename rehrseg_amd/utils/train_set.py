"""Training-patch data sets with the volumes resident in HBM, under the reference's import path
(`utils.train_set`).

The reference cuts every patch on the host with numpy and hands it to the loader
(utils/train_set.py:21-160 TrainSetMultipleSegSREfficient, :162-223 TrainSetMultipleSegSR, :226-434 TrainSetMultiple;
built by train_all.py:291, :368, :501).  Here the volumes of all subjects are uploaded once; `__getitem__` draws the
SAME numbers from Python's `random` in the SAME order as the reference, turns the reference's chain of numpy
operations (transpose, crop, constant pad, flip, every-k-th slice, permute) into one strided-gather descriptor per
output (`View`, integer bookkeeping only) and launches `rehr_patch_gather` / `rehr_axis_resample_f32`
(csrc/patch_feed.hip).  `batch(indices)` does the same for a whole batch with one launch per stage.

What is pinned by fixtures generated from the reference's own classes (tools/gen_golden_feed.py): the random-number
protocol, crop / pad / flip / transposition / zero-slice / permute logic of all three `__getitem__`s.  What is not
(packages absent offline, restated from their published behaviour, "parity unpinned"): `resize.pytorch.resize`
(`resize_axis` below), `degrade.select_kernel` (utils/blur_kernel_ops.py of this package), the batchgenerators
augmentation chain of stage 2 (`train_transform`: pass a callable, default identity), h5py / nibabel containers
(`.npz` files with the reference's H5 keys are read instead; h5py is used when importable).
"""
import math
import os
import random

import numpy as np
import torch

from .. import hip_backend as hb
from .pad import get_pads
from .seg_utils import zscore_normalization


# ----------------------------------------------------------------------------- strided views (host integers only)
class View:
    """A chain of numpy-style operations over one volume, kept as (shape, stride, base, valid box): element `o` of the
    view is volume[base + sum_k o_k * stride_k] when lo_k <= o_k < hi_k on every axis and 0 (the constant pad)
    otherwise -- exactly one `rehr_patch_item`."""

    def __init__(self, shape, stride=None, base=0, lo=None, hi=None):
        self.shape = [int(s) for s in shape]
        if stride is None:
            stride, acc = [], 1
            for s in reversed(self.shape):
                stride.insert(0, acc)
                acc *= s
        self.stride = [int(s) for s in stride]
        self.base = int(base)
        self.lo = [0] * len(self.shape) if lo is None else [int(v) for v in lo]
        self.hi = list(self.shape) if hi is None else [int(v) for v in hi]
        self.dead = False  # a squeezed-away axis consisted of padding only: every element reads 0

    def _new(self, order):
        v = View([self.shape[k] for k in order], [self.stride[k] for k in order], self.base,
                 [self.lo[k] for k in order], [self.hi[k] for k in order])
        v.dead = self.dead
        return v

    def copy(self):
        return self._new(range(len(self.shape)))

    def transpose(self, *order):
        order = order[0] if len(order) == 1 and not isinstance(order[0], int) else order
        assert sorted(order) == list(range(len(self.shape)))
        return self._new(order)

    permute = transpose

    def slice(self, axis, start, stop):
        """v[start:stop] along `axis` with Python's clamping (0 <= start; stop may exceed the extent)."""
        v = self.copy()
        n = v.shape[axis]
        start, stop = min(max(int(start), 0), n), min(max(int(stop), 0), n)
        stop = max(stop, start)
        v.base += start * v.stride[axis]
        v.shape[axis] = stop - start
        v.lo[axis] = min(max(v.lo[axis] - start, 0), stop - start)
        v.hi[axis] = min(max(v.hi[axis] - start, 0), stop - start)
        return v

    def crop(self, starts, sizes):
        v = self
        for k, (a, n) in enumerate(zip(starts, sizes)):
            v = v.slice(k, a, a + n)
        return v

    def pad(self, axis, before, after):
        v = self.copy()
        v.base -= before * v.stride[axis]
        v.shape[axis] += before + after
        v.lo[axis] += before
        v.hi[axis] += before
        return v

    def target_pad(self, target_dims):
        """utils/pad.py:14-21 with mode="constant": centred, the odd voxel behind."""
        v = self
        for k, (t, d) in enumerate(zip(target_dims, self.shape)):
            b, a = get_pads(t, d)
            v = v.pad(k, b, a)
        return v

    def flip(self, axis):
        v = self.copy()
        n = v.shape[axis]
        v.base += (n - 1) * v.stride[axis]
        v.stride[axis] = -v.stride[axis]
        v.lo[axis], v.hi[axis] = n - self.hi[axis], n - self.lo[axis]
        return v

    def step(self, axis, k, start=0):
        """v[start::k] along `axis`."""
        v = self.copy()
        n = v.shape[axis]
        cnt = lambda e: max(0, -((start - e) // k))  # number of i >= 0 with start + k*i < e  # noqa: E731
        v.base += start * v.stride[axis]
        v.stride[axis] *= k
        v.shape[axis] = cnt(n)
        v.lo[axis], v.hi[axis] = cnt(self.lo[axis]), cnt(self.hi[axis])
        return v

    def zero(self, axis, index):
        """v[..., index, ...] = 0 for the first (index 0) or last (index -1) position of `axis`."""
        v = self.copy()
        n = v.shape[axis]
        if index == 0:
            v.lo[axis] = max(v.lo[axis], min(1, n))
        elif index == -1:
            v.hi[axis] = min(v.hi[axis], max(n - 1, 0))
        else:
            raise ValueError("only the first or the last position can be blanked")
        v.hi[axis] = max(v.hi[axis], v.lo[axis])
        return v

    def expand(self, axis):
        v = self.copy()
        for lst, val in ((v.shape, 1), (v.stride, 0), (v.lo, 0), (v.hi, 1)):
            lst.insert(axis, val)
        return v

    def squeeze(self, axis):
        """torch's x.squeeze(axis): a no-op unless the extent is 1."""
        if self.shape[axis] != 1:
            return self
        v = self.copy()
        v.dead = v.dead or v.lo[axis] >= v.hi[axis]
        for lst in (v.shape, v.stride, v.lo, v.hi):
            del lst[axis]
        return v

    def item(self, src):
        """(src, base, stride[4], lo[4], hi[4]) and the 4-axis extent for hip_backend.patch_gather: extent-1 axes
        are dropped, the rest right-aligned into four axes."""
        keep = [k for k, s in enumerate(self.shape) if s != 1]
        empty = self.dead or any(self.lo[k] >= self.hi[k] for k in range(len(self.shape)))
        if len(keep) > 4:
            raise ValueError("a patch has at most four non-trivial axes")
        base = self.base
        padn = 4 - len(keep)
        dims = [1] * padn + [self.shape[k] for k in keep]
        stride = [0] * padn + [self.stride[k] for k in keep]
        lo = [0] * padn + [self.lo[k] for k in keep]
        hi = [1] * padn + [self.hi[k] for k in keep]
        if empty:
            lo, hi = [0] * 4, [0] * 4
        return (src, base, stride, lo, hi), dims


def _gather(views, srcs, scale=1.0, bias=0.0):
    """One launch for a batch of equally shaped views; returns float32 [len(views), *shape]."""
    shape = views[0].shape
    items, dims = [], None
    for v, s in zip(views, srcs):
        if v.shape != shape:
            raise ValueError(f"patches of one batch must have one shape, got {v.shape} and {shape}")
        it, dims = v.item(s)
        items.append(it)
    out = hb.patch_gather(items, dims, scale, bias)
    return out.view((len(views),) + tuple(shape))


# ----------------------------------------------------------------------------- 1-D resampling / blur as tap tables
def _cubic_weights(t, a=-0.75):
    """Keys' cubic convolution weights of the 4 neighbours at fractional offset t (torch's bicubic, a = -0.75)."""
    def k(x):
        x = abs(x)
        if x <= 1:
            return ((a + 2) * x - (a + 3)) * x * x + 1
        if x < 2:
            return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
        return 0.0
    return [k(t + 1), k(t), k(1 - t), k(2 - t)]


def resize_taps(n_in, dx, order):
    """Tap table of `resize(x, (dx, ...), order)` along one axis (resize.pytorch of the iacl `resize` package, absent
    offline -- restated from its documented behaviour, PARITY UNPINNED): same field of view, n_out = round(n_in / dx)
    samples at (dx - 1) / 2 + j * dx; order 0 nearest (ties to even, as torch's grid sampler), 1 linear, 3 cubic
    convolution; positions beyond the ends reflect about the edge samples.  dx == 1 is the identity."""
    if order not in (0, 1, 3):
        raise ValueError("order 0, 1 or 3")
    n_out = int(round(n_in / dx))
    taps = {0: 1, 1: 2, 3: 4}[order]
    idx = np.zeros((n_out, taps), np.int32)
    w = np.zeros((n_out, taps), np.float32)

    def refl(i):
        if n_in == 1:
            return 0
        period = 2 * (n_in - 1)
        i = abs(i) % period
        return period - i if i >= n_in else i

    for j in range(n_out):
        p = (dx - 1) / 2.0 + j * dx
        if order == 0:
            r = math.floor(p + 0.5)
            if p + 0.5 == r and r % 2:  # tie: to even
                r -= 1
            idx[j, 0], w[j, 0] = refl(r), 1.0
        elif order == 1:
            f = math.floor(p)
            idx[j], w[j] = [refl(f), refl(f + 1)], [1 - (p - f), p - f]
        else:
            f = math.floor(p)
            idx[j], w[j] = [refl(f - 1), refl(f), refl(f + 1), refl(f + 2)], _cubic_weights(p - f)
    return idx, w


def blur_taps(n, kernel):
    """F.conv2d(x, kernel[None, None, :, None], padding="same") along one axis (utils/train_set.py:306-318) as a tap
    table: cross-correlation, zero padding, (L - 1) // 2 samples in front."""
    kernel = np.asarray(kernel, np.float32).reshape(-1)
    L = kernel.shape[0]
    left = (L - 1) // 2
    idx = np.full((n, L), -1, np.int32)
    w = np.zeros((n, L), np.float32)
    for j in range(n):
        for t in range(L):
            s = j + t - left
            if 0 <= s < n:
                idx[j, t], w[j, t] = s, kernel[t]
    return idx, w


def _check_taps(idx, n_in):
    if idx.size and int(idx.max()) >= n_in:
        raise ValueError("tap index beyond the axis")


def _resample(x, axis, table):
    idx, w = table
    _check_taps(idx, x.shape[axis])
    return hb.axis_resample(x.contiguous(), axis, torch.from_numpy(idx).to(x.device), torch.from_numpy(w).to(x.device),
                            validated=True)


_tap_cache = {}   # (n_in, dx, order, device) -> (idx, w) on the device: the tables of a training run never change


def resize_axis(x, axis, dx, order):
    """`resize` along one axis of a device tensor (see resize_taps; unpinned)."""
    if dx == 1:
        return x
    key = (x.shape[axis], float(dx), order, x.device)
    if key not in _tap_cache:
        idx, w = resize_taps(x.shape[axis], dx, order)
        _check_taps(idx, x.shape[axis])
        _tap_cache[key] = (torch.from_numpy(idx).to(x.device), torch.from_numpy(w).to(x.device))
    return hb.axis_resample(x.contiguous(), axis, *_tap_cache[key], validated=True)


# ----------------------------------------------------------------------------- containers
def _read_container(path):
    """The reference stores its merged data sets as H5 (train_all.py:34-62; h5py absent offline): `.npz` files with the
    same keys are read here, `.h5` through h5py when it is importable."""
    if path.endswith(".npz"):
        with np.load(path) as f:
            return {k: f[k] for k in f.files}
    if path.endswith(".h5"):
        try:
            import h5py
        except ImportError as e:
            raise ImportError("reading .h5 data sets needs h5py (absent here): store the same keys in a .npz") from e
        with h5py.File(path, "r") as f:
            return {k: f[k][:] for k in f.keys()}
    raise ValueError(f"unsupported data set container {path} (.npz or .h5)")


def _dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


def _flips(random_flip, n=3):
    """The reference draws one number per axis, in axis order, only when flipping is enabled."""
    return [random_flip and random.random() < 0.5 for _ in range(n)] if random_flip else [False] * n


class _DeviceSet(torch.utils.data.Dataset):
    device = None

    def _check_device(self, device):
        device = torch.device("cuda" if device is None else device)
        if device.type != "cuda":
            raise hb.L.RehrsegHipError("the patch feed keeps its volumes on the GPU (no CPU fallback)")
        return device

    def batch(self, indices):
        """The items `indices` (drawn in this order from `random`, as a loader with num_workers=0 would) stacked along a
        new leading axis, with one launch per stage instead of one per item."""
        return self._run([self._plan(i) for i in indices])

    def __getitem__(self, i):
        if torch.utils.data.get_worker_info() is not None:
            # the reference's stage-2 loader forks 4 workers (train_all.py:502-509); these data sets live on the GPU
            # and launch kernels, which a forked worker must not do
            raise hb.L.RehrsegHipError(
                "rehrseg_amd data sets keep their volumes in HBM and cut patches with HIP kernels: use them from the "
                "training process (DataLoader(num_workers=0, pin_memory=False), or ds.batch(indices)), not from "
                "DataLoader worker processes")
        out = self._run([self._plan(i)])
        return tuple(o[0] if torch.is_tensor(o) else o for o in out)


# ----------------------------------------------------------------------------- stage 2: image / label / uncertainty patches
class TrainSetMultipleSegSREfficient(_DeviceSet):
    """utils/train_set.py:21-160.  `volumes` (one dict per subject with 'img', 'seg' and, with `uncertainty`,
    'uncertainty' arrays of shape (x, y, z)) replaces the H5 files when given.  `train_transform` stands where the
    reference calls its batchgenerators chain (:64-84, absent offline): a callable over the keyword tensors
    data / seg / seg_sr / uncertainty, default identity."""

    def __init__(self, image_path, split_subjects, slice_thickness, target_thickness, patch_size_ori, target_patch_size,
                 random_flip=False, uncertainty=False, preload=True, norm=True, device=None, volumes=None,
                 train_transform=None):
        self.image_path, self.patch_size = image_path, list(patch_size_ori)
        self.slice_thickness, self.target_thickness = slice_thickness, target_thickness
        self.separation = int(slice_thickness / target_thickness)
        self.random_flip, self.uncertainty, self.norm = random_flip, uncertainty, norm
        self.target_patch_size = target_patch_size
        self.train_transform = train_transform
        self.device = self._check_device(device)
        self.imgs, self.labels, self.uncertainties = [], [], []
        if volumes is None:
            volumes = [_read_container(self._subject_file(image_path, s)) for s in split_subjects]
        for v in volumes:
            img = np.asarray(v["img"])
            if norm:  # the reference normalises the whole volume on every access (:104-105): once is the same numbers
                img = zscore_normalization(img.copy())
            self.imgs.append(_dev(img, self.device, torch.float32))
            self.labels.append(_dev(v["seg"], self.device, torch.uint8))
            self.uncertainties.append(_dev(v["uncertainty"], self.device, torch.uint8) if uncertainty else None)
        print("Total subjects", len(self.imgs))

    @staticmethod
    def _subject_file(image_path, subject):
        for ext in (".h5", ".npz"):
            p = os.path.join(image_path, subject + "_0000" + ext)
            if os.path.exists(p):
                return p
        raise FileNotFoundError(os.path.join(image_path, subject + "_0000.h5"))

    def __len__(self):
        return len(self.imgs)

    def _plan(self, i):
        shape = tuple(self.imgs[i].shape)
        ps, sep = self.patch_size, self.separation
        ext = (ps[0], ps[1], ps[2] * sep)
        x0 = random.randint(0, max(shape[0] - ext[0], 0))
        y0 = random.randint(0, max(shape[1] - ext[1], 0))
        z0 = random.randint(0, max(shape[2] - ext[2], 0))
        v = View(shape).crop((x0, y0, z0), ext)
        v = v.target_pad([max(s, p) for s, p in zip(v.shape, ext)])
        for axis, f in enumerate(_flips(self.random_flip)):
            if f:
                v = v.flip(axis)
        to5 = lambda u: u.transpose(2, 1, 0).expand(0).expand(0)  # noqa: E731  (1, 1, z, y, x)
        return i, to5(v.step(2, sep)), to5(v)

    def _run(self, plans):
        ids = [p[0] for p in plans]
        lr = [p[1] for p in plans]
        hr = [p[2] for p in plans]
        data = {"data": _gather(lr, [self.imgs[i] for i in ids]),
                "seg": _gather(lr, [self.labels[i] for i in ids]),
                "seg_sr": _gather(hr, [self.labels[i] for i in ids])}
        if self.uncertainty:  # 1 - u / 255 * 0.99, the pad included (:147)
            data["uncertainty"] = _gather(lr, [self.uncertainties[i] for i in ids], -0.99 / 255.0, 1.0)
        if self.train_transform is not None:
            data = self.train_transform(**data)
        # the per-item tensors are (1, 1, z, y, x); .squeeze(0) leaves (1, z, y, x) (:149-158)
        sq = lambda t: t.squeeze(1)  # noqa: E731
        unc = sq(data["uncertainty"]) if self.uncertainty else 0
        return sq(data["data"]), sq(data["seg"]), sq(data["seg_sr"]), unc


# ----------------------------------------------------------------------------- image + label channel patches (NIfTI merge)
class TrainSetMultipleSegSR(_DeviceSet):
    """utils/train_set.py:162-223.  `volumes`: one (x, y, z, 2) array per subject (image, label), as `parse_image`
    returns for the merged NIfTI files (nibabel absent offline: arrays only)."""

    def __init__(self, image_path, split_subjects, slice_thickness, target_thickness, patch_size, random_flip=False,
                 device=None, volumes=None):
        if len(patch_size) == 2:
            patch_size = (*patch_size, 1)
        self.patch_size, self.random_flip, self.split_subjects = patch_size, random_flip, split_subjects
        self.device = self._check_device(device)
        if volumes is None:
            raise ImportError("reading the merged .nii.gz needs nibabel (absent here): pass volumes=[(x, y, z, 2) arrays]")
        self.imgs, self.labels = [], []
        for each_subject, image in zip(split_subjects, volumes):
            image = np.asarray(image).squeeze()
            if image.ndim == 3:
                image = image[..., np.newaxis]
            # the reference pads (x, y, z, c) towards a five-entry target (:186-190): zip stops at four axes, the
            # channel axis is padded up to image.shape[3] (no change)
            target = [max(s, p) for s, p in zip(image.shape[:3], self.patch_size[:3])] + [image.shape[3]]
            pads = [get_pads(t, d) for t, d in zip(target, image.shape)]
            image = np.pad(image, pads, mode="constant")
            print(each_subject, "image shape", image.shape)
            self.imgs.append(_dev(image[..., :1], self.device, torch.float32))
            self.labels.append(_dev(image[..., 1:].astype("uint8"), self.device, torch.uint8))
        print("Total subjects", len(self.imgs))

    def __len__(self):
        return len(self.imgs)

    def _plan(self, i):
        shape = tuple(self.imgs[i].shape)
        ps = self.patch_size
        x0 = random.randint(0, shape[0] - ps[0])
        y0 = random.randint(0, shape[1] - ps[1])
        z0 = random.randint(0, shape[2] - ps[2])
        v = View(shape).crop((x0, y0, z0), ps[:3])
        for axis, f in enumerate(_flips(self.random_flip)):
            if f:
                v = v.flip(axis)
        return i, v.transpose(3, 2, 1, 0)

    def _run(self, plans):
        ids, views = [p[0] for p in plans], [p[1] for p in plans]
        return _gather(views, [self.imgs[i] for i in ids]), _gather(views, [self.labels[i] for i in ids])


# ----------------------------------------------------------------------------- stage 1: (LR, HR) pairs for the SR network
class TrainSetMultiple(_DeviceSet):
    """utils/train_set.py:226-434.  Per subject: img_hr (x, y, z, 1) float, label_hr (x, y, z, 1) uint8 and, with `blur`,
    the in-plane slice-profile blurred copies image_x_rgb (z, 1, x, y) / image_y_rgb (z, 1, y, x) -- read from the
    merged container, or built here on the device from `volumes=[(x, y, z, 2) arrays]` and `blur_kernel` (:295-318)."""

    def __init__(self, image_path, split_subjects, slice_thickness, target_thickness, blur_kernel_fpath, blur_kernel_name,
                 patch_size, random_flip, device, preload=True, blur=True, nnunet_transform=False, norm=True, volumes=None,
                 blur_kernel=None):
        if len(patch_size) == 2:
            patch_size = (*patch_size, 1)
        self.patch_size, self.random_flip, self.blur = patch_size, random_flip, blur
        self.device = self._check_device(device)
        self.all_subjects = split_subjects
        self.slice_thickness, self.target_thickness = slice_thickness, target_thickness
        self.slice_separation = float(slice_thickness / target_thickness)
        if nnunet_transform:
            raise NotImplementedError("the nnU-Net augmentation chain (batchgenerators, :260-276) is absent offline")
        self.imgs_hr, self.labels_hr, self.imgs_filtered_x, self.imgs_filtered_y = [], [], [], []
        if volumes is None:
            names = os.listdir(image_path)
            volumes = [_read_container(os.path.join(image_path, [x for x in names if s in x][0])) for s in split_subjects]
        for s, v in zip(split_subjects, volumes):
            if isinstance(v, dict):
                img_hr, label_hr = np.asarray(v["img_hr"]), np.asarray(v["label_hr"])
                fx = _dev(v["image_x_rgb"], self.device, torch.float32) if blur else None
                fy = _dev(v["image_y_rgb"], self.device, torch.float32) if blur else None
                img_hr = _dev(img_hr, self.device, torch.float32)
            else:
                image = np.asarray(v).squeeze()
                if image.ndim == 3:
                    image = image[..., np.newaxis]
                img_hr, label_hr = _dev(image[..., :1], self.device, torch.float32), image[..., 1:].astype("uint8")
                fx = fy = None
                if blur:
                    if blur_kernel is None:
                        from .blur_kernel_ops import parse_kernel
                        from .parse_image_file import blur_fwhm_voxels
                        blur_kernel = parse_kernel(blur_kernel_fpath, blur_kernel_name,
                                                   blur_fwhm_voxels(slice_thickness, target_thickness))
                    k = np.asarray(blur_kernel, np.float32).reshape(-1)
                    vol = img_hr[..., 0]  # (x, y, z)
                    fx = _resample(vol, 0, blur_taps(vol.shape[0], k)).permute(2, 0, 1).unsqueeze(1).contiguous()
                    fy = _resample(vol, 1, blur_taps(vol.shape[1], k)).permute(2, 1, 0).unsqueeze(1).contiguous()
            print(s, "image shape", tuple(img_hr.shape))
            self.imgs_hr.append(img_hr)
            self.labels_hr.append(_dev(label_hr, self.device, torch.uint8))
            self.imgs_filtered_x.append(fx)
            self.imgs_filtered_y.append(fy)

    def __len__(self):
        return len(self.all_subjects)

    def _plan(self, i):
        ps = self.patch_size
        hr = View(self.imgs_hr[i].shape)          # (x, y, z, 1)
        swap = random.random() < 0.5              # one draw with or without blur (:331-343)
        if swap:
            hr = hr.transpose(1, 0, 2, 3)
        lr_src = None
        if self.blur:
            lr_src = self.imgs_filtered_y[i] if swap else self.imgs_filtered_x[i]
        x0 = random.randint(0, max(hr.shape[0] - ps[0], 0))
        y0 = random.randint(0, max(hr.shape[1] - ps[1], 0))
        z0 = random.randint(0, max(hr.shape[2] - ps[2], 0))
        hr = hr.crop((x0, y0, z0, 0), (ps[0], ps[1], ps[2], hr.shape[3])).transpose(2, 3, 0, 1)  # z, channel, x, y
        target = [max(s, p) for s, p in zip(hr.shape, (ps[2], 1, ps[0], ps[0]))]  # ps[0] twice, as the reference (:355)
        hr = hr.target_pad(target)
        if self.blur:
            lr = View(lr_src.shape).crop((z0, 0, x0, y0), (ps[2], lr_src.shape[1], ps[0], ps[1])).target_pad(target)
        else:
            lr = hr
        # the tail of __getitem__ (:406-434) in the reference's draw order; applied to both tensors after the resize
        deep = hr.shape[0] > 1
        zero_first = deep and random.random() < 0.1
        zero_last = deep and random.random() < 0.1
        flips = _flips(self.random_flip)
        to_xyz = random.random() < 0.5
        return dict(i=i, swap=swap, hr=hr, lr=lr, lr_src=lr_src, zero_first=zero_first, zero_last=zero_last, flips=flips,
                    to_xyz=to_xyz)

    @staticmethod
    def _tail(v, p, is_lr):
        """(z, c, x', y) -> permute(1, 2, 0, 3) -> zero slices (LR only) -> flips -> permute / squeeze (:406-434)."""
        v = v.permute(1, 2, 0, 3)  # channel, x, z, y
        if is_lr:
            if p["zero_first"]:
                v = v.zero(1, 0)
            if p["zero_last"]:
                v = v.zero(1, -1)
        for axis, f in zip((1, 2, 3), p["flips"]):
            if f:
                v = v.flip(axis)
        if p["to_xyz"]:
            return v.permute(0, 1, 3, 2).squeeze(3)
        return v.squeeze(2)

    def _run(self, plans):
        ids = [p["i"] for p in plans]
        sep = self.slice_separation
        # HR pair: image and label channels gathered straight into their final orientation
        hr_views = [self._tail(p["hr"], p, False) for p in plans]
        img_hr = _gather(hr_views, [self.imgs_hr[i] for i in ids])
        lab_hr = _gather(hr_views, [self.labels_hr[i] for i in ids])
        out_hr = torch.cat((img_hr, lab_hr), dim=1)
        # LR pair: blurred patch (z, 1, x, y) -> cubic down-sampling along x -> final orientation; the order-0 label is
        # a strided gather of the label volume itself
        src = [p["lr_src"] if self.blur else self.imgs_hr[p["i"]] for p in plans]
        patch = _gather([p["lr"] for p in plans], src)                       # (B, z, 1, x, y)
        low = resize_axis(patch, 3, sep, 3)                                  # (B, z, 1, x / sep, y)
        idx0, _ = resize_taps(plans[0]["hr"].shape[2], sep, 0)
        lr_views, lab_views = [], []
        for b, p in enumerate(plans):
            lr_views.append(self._tail(View(low.shape[1:]), p, True))
            lab_views.append(self._tail(_take(p["hr"], 2, idx0[:, 0]), p, True))
        # every item reads its own slab of `low`
        slab = low[0].numel()
        items_src = [low] * len(plans)
        for b, v in enumerate(lr_views):
            v.base += b * slab
        img_lr = _gather(lr_views, items_src)
        lab_lr = _gather(lab_views, [self.labels_hr[i] for i in ids])
        return torch.cat((img_lr, lab_lr), dim=1), out_hr


def _take(v, axis, idx):
    """v[..., idx, ...] for an arithmetic index sequence (the order-0 resize picks every dx-th sample)."""
    idx = [int(k) for k in idx]
    if len(idx) == 0:
        return v.slice(axis, 0, 0)
    if len(idx) == 1:
        return v.slice(axis, idx[0], idx[0] + 1)
    k = idx[1] - idx[0]
    if k <= 0 or any(b - a != k for a, b in zip(idx, idx[1:])):
        raise ValueError("nearest-neighbour indices are not an arithmetic sequence (non-integer slice separation)")
    return v.slice(axis, 0, idx[-1] + 1).step(axis, k, idx[0])
