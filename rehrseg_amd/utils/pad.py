"""Centred padding / cropping helpers of the reference's data path, under its import path (`utils.pad`): same names
and results as utils/pad.py:5-34 (pinned by tests/golden/feed_misc.json and the data-set fixtures).  The patch feed
applies the same pad amounts inside its gather descriptors (`View.target_pad` in utils/train_set.py of this package)."""
import numpy as np
import torch


def get_pads(target_dim, d):
    """(before, after): what is missing up to `target_dim`, split in the middle with the odd voxel behind; (0, 0) when
    nothing is missing."""
    missing = max(int(target_dim) - int(d), 0)
    return missing // 2, missing - missing // 2


def target_pad(img, target_dims, mode="reflect"):
    """`img` (numpy array or CPU tensor) padded out to `target_dims` (axes beyond len(target_dims) stay as they are);
    returns (padded, per-axis (before, after))."""
    as_tensor = isinstance(img, torch.Tensor)
    arr = img.numpy() if as_tensor else img
    widths = tuple(get_pads(t, n) for t, n in zip(target_dims, arr.shape))
    out = np.pad(arr, widths, mode=mode)
    return (torch.Tensor(out) if as_tensor else out), widths


def format_pads(pads):
    """The slice that removes a (before, after) pair again; zero amounts become open ends."""
    before, after = pads
    return slice(before or None, -after if after else None)


def crop(img, pads):
    return img[tuple(format_pads(p) for p in pads)]
