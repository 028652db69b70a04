"""Slice-profile blur kernel helpers under the reference's import path (`utils.blur_kernel_ops`, ref :7-36).

`degrade.select_kernel` (iacl `degrade` package) is absent offline: its Gaussian branch is restated here from the
package's published behaviour -- a `window_size`-tap Gaussian window with sigma = fwhm / (2 sqrt(2 ln 2)) -- PARITY
UNPINNED; the other kernel types it offers (RF-pulse profiles) are not available.  `calc_extended_patch_size` is pinned
by a fixture generated from the reference (tests/golden/feed_misc.json)."""
from math import ceil

import numpy as np
import torch


def select_kernel(window_size, kernel_type="gaussian", fwhm=None):
    if kernel_type != "gaussian":
        raise NotImplementedError(f"blur kernel type {kernel_type!r}: only 'gaussian' is restated (degrade is absent)")
    sigma = fwhm / (2.0 * np.sqrt(2.0 * np.log(2.0)))
    n = np.arange(window_size, dtype=np.float64) - (window_size - 1) / 2.0
    return np.exp(-0.5 * (n / sigma) ** 2)


def parse_kernel(blur_kernel_file, blur_kernel_type, blur_fwhm):
    """(1, 1, L, 1) float32 kernel, normalised to sum 1 (ref :7-18)."""
    if blur_kernel_file is not None:
        blur_kernel = np.load(blur_kernel_file)
    else:
        window_size = int(2 * round(blur_fwhm) + 1)
        blur_kernel = select_kernel(window_size, blur_kernel_type, fwhm=blur_fwhm)
    blur_kernel = blur_kernel / blur_kernel.sum()
    blur_kernel = blur_kernel.squeeze()[None, None, :, None]
    return torch.from_numpy(blur_kernel).float()


def calc_extended_patch_size(blur_kernel, patch_size):
    """Patch size grown by the blur support on every non-singleton axis, and the slices that crop it back (ref :21-36)."""
    L = blur_kernel.shape[0]
    ext_patch_size = [p + 2 * ceil(L / 2) if p != 1 else p for p in patch_size]
    ext_patch_crop = [(e - p) // 2 for e, p in zip(ext_patch_size, patch_size)]
    ext_patch_crop = tuple([slice(d, -d) for d in ext_patch_crop if d != 0])
    return ext_patch_size, ext_patch_crop
