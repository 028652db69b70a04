"""Centred padding / cropping helpers of the reference's data path, under its import path (`utils.pad`): same names
and results as utils/pad.py:5-34 (host-side numpy, as in the reference; the patch feed applies the same pad amounts
inside its gather descriptors, utils/train_set.py of this package)."""
import numpy as np
import torch


def get_pads(target_dim, d):
    """(before, after) so that d + before + after == target_dim, the odd voxel behind (ref :5-11)."""
    if target_dim <= d:
        return 0, 0
    p = (target_dim - d) // 2
    return p, target_dim - d - p


def target_pad(img, target_dims, mode="reflect"):
    """Pad `img` (numpy array or CPU tensor) out to `target_dims`; returns (padded, pads) (ref :14-21)."""
    pads = tuple(get_pads(t, d) for t, d in zip(target_dims, img.shape))
    if isinstance(img, torch.Tensor):
        return torch.Tensor(np.pad(img.numpy(), pads, mode=mode)), pads
    return np.pad(img, pads, mode=mode), pads


def format_pads(pads):
    """A (before, after) pair as the slice that removes it; 0 becomes an open end (ref :24-28)."""
    return slice(pads[0] if pads[0] != 0 else None, -pads[1] if pads[1] != 0 else None)


def crop(img, pads):
    return img[tuple(map(format_pads, pads))]
