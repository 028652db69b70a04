"""Intensity normalisation and orientation helpers under the reference's import path (`utils.parse_image_file`,
ref :7-21, :100-131).  The file readers of the reference (`parse_image`, `LazyHDF5File`, :24-96) sit on nibabel / h5py,
both absent offline: utils/train_set.py of this package takes arrays or `.npz` containers instead.
`blur_fwhm_voxels` restates the two `degrade` helpers `parse_image` calls (:85; package absent, PARITY UNPINNED)."""
import numpy as np


def normalize(x, a=-1, b=1):
    orig_min, orig_max = x.min(), x.max()
    return a + (x - orig_min) * (b - a) / (orig_max - orig_min), orig_min, orig_max


def inv_normalize(x, orig_min, orig_max, a=-1, b=1):
    tmp = (x - a) * (orig_max - orig_min) / (b - a)
    tmp += orig_min
    return tmp


def blur_fwhm_voxels(slice_thickness, target_thickness):
    """FWHM (in target voxels) of the blur that turns a `target_thickness` profile into a `slice_thickness` one:
    sqrt(slice^2 - target^2) / target (degrade.fwhm_needed + fwhm_units_to_voxel_space)."""
    return float(np.sqrt(slice_thickness ** 2 - target_thickness ** 2) / target_thickness)


_TO_Z = {0: (2, 0, 1, 3), 1: (1, 2, 0, 3)}   # axis order that moves the low-resolution axis where the reference wants it


def lr_axis_to_z(img, lr_axis):
    """(ref :100-113) a trailing singleton fifth axis is squeezed away first."""
    if img.ndim == 5:
        img = np.squeeze(img)
    return img.transpose(_TO_Z[lr_axis]) if lr_axis in _TO_Z else img


def z_axis_to_lr_axis(img, lr_axis):
    """(ref :117-131) the reference applies the same permutation in this direction."""
    if img.ndim == 5:
        img = np.squeeze(img, axis=4)
    return img.transpose(_TO_Z[lr_axis]) if lr_axis in _TO_Z else img
