"""Slice-profile blur kernel helpers under the reference's import path (`utils.blur_kernel_ops`, ref :7-36).

`degrade.select_kernel` (iacl `degrade` package) is absent offline: its Gaussian branch is restated here from the
package's published behaviour -- a `window_size`-tap Gaussian window with sigma = fwhm / (2 sqrt(2 ln 2)) -- PARITY
UNPINNED; the other kernel types it offers (RF-pulse profiles) are not available.  `calc_extended_patch_size` is pinned
by a fixture generated from the reference (tests/golden/feed_misc.json)."""
import math

import numpy as np
import torch

_FWHM_TO_SIGMA = 1.0 / (2.0 * math.sqrt(2.0 * math.log(2.0)))


def select_kernel(window_size, kernel_type="gaussian", fwhm=None):
    if kernel_type != "gaussian":
        raise NotImplementedError(f"blur kernel type {kernel_type!r}: only 'gaussian' is restated (degrade is absent)")
    taps = np.arange(window_size, dtype=np.float64) - 0.5 * (window_size - 1)
    return np.exp(-0.5 * np.square(taps / (fwhm * _FWHM_TO_SIGMA)))


def parse_kernel(blur_kernel_file, blur_kernel_type, blur_fwhm):
    """The slice profile as a (1, 1, L, 1) float32 conv2d weight of unit sum: read from `blur_kernel_file` (.npy) when
    given, else a window of 2 round(fwhm) + 1 taps of the named type."""
    if blur_kernel_file is None:
        profile = select_kernel(2 * int(round(blur_fwhm)) + 1, blur_kernel_type, fwhm=blur_fwhm)
    else:
        profile = np.load(blur_kernel_file)
    profile = np.asarray(profile, dtype=np.float64).squeeze()
    profile = profile / profile.sum()
    return torch.from_numpy(profile.reshape(1, 1, -1, 1)).float()


def calc_extended_patch_size(blur_kernel, patch_size):
    """Every non-singleton patch axis grown by 2 ceil(L / 2) voxels (L = blur support) so that the blur's border effects
    can be cut off again, and the slices that do the cutting (one per grown axis)."""
    margin = math.ceil(blur_kernel.shape[0] / 2)
    grown = [p if p == 1 else p + 2 * margin for p in patch_size]
    cuts = tuple(slice((g - p) // 2, -((g - p) // 2)) for g, p in zip(grown, patch_size) if g != p)
    return grown, cuts
