"""Shared by tools/gen_golden_feed.py (which runs the reference) and the feed tests: the case tables, the seeded
volumes, and a numpy emulation of the two feed kernels' C-ABI semantics (include/rehrseg_hip.h: rehr_patch_gather,
rehr_axis_resample_f32) that the CPU tests bind in place of the GPU launches to check the host-side descriptors."""
import numpy as np
import torch

KERNEL = np.array([0.05, 0.2, 0.5, 0.2, 0.05], np.float32)
MULTI_CASES = {
    # name: (volume shapes, patch_size, slice thickness / target, blur, random_flip, seed, draws)
    "multi_2d_blur": ([(20, 24, 5), (18, 17, 4)], (16, 16, 1), 4.0, True, True, 11, 14),
    "multi_3d_blur": ([(20, 12, 9), (14, 9, 7)], (16, 8, 8), 4.0, True, True, 12, 16),
    "multi_3d_noblur": ([(20, 12, 9)], (16, 8, 8), 2.0, False, False, 13, 8),
    "multi_3d_cube": ([(20, 12, 9), (7, 9, 11)], (8, 8, 8), 2.0, True, True, 14, 12),
}
BATCHABLE = ("multi_2d_blur", "multi_3d_cube")  # every draw has one shape (the loader's collate needs that too)
SEGSR_CASES = {"segsr": ([(14, 12, 9), (6, 13, 5)], (8, 8, 4), True, 21, 8)}
EFF_CASES = {
    # name: (volume shapes, patch_size_ori, separation, uncertainty, random_flip, norm, seed, draws)
    "eff_unc": ([(14, 12, 26), (9, 13, 10)], (10, 10, 3), 4, True, True, True, 31, 8),
    "eff_plain": ([(14, 12, 26)], (10, 10, 3), 2, False, False, False, 32, 4),
}


def volumes_multi(seed, shapes):
    """(x, y, z, 2) image + binary label volumes."""
    rng = np.random.RandomState(seed)
    vols = []
    for s in shapes:
        img = rng.rand(*s).astype(np.float32)
        lab = (rng.rand(*s) > 0.6).astype(np.float32)
        vols.append(np.stack((img, lab), axis=-1))
    return vols


def volumes_seg(seed, shapes):
    rng = np.random.RandomState(seed)
    return [dict(img=(rng.rand(*s) * 100).astype(np.float32), seg=(rng.rand(*s) > 0.5).astype(np.uint8),
                 uncertainty=rng.randint(0, 256, size=s).astype(np.uint8)) for s in shapes]


# ----------------------------------------------------------------------------- kernel semantics in numpy (tests only)
def emu_patch_gather(items, dims, scale=1.0, bias=0.0):
    out = np.zeros((len(items),) + tuple(dims), np.float32)
    o = np.indices(dims)
    for i, (src, base, stride, lo, hi) in enumerate(items):
        flat = src.detach().cpu().numpy().reshape(-1)
        off = np.full(dims, int(base), np.int64)
        ok = np.ones(dims, bool)
        for k in range(4):
            off += o[k] * int(stride[k])
            ok &= (o[k] >= lo[k]) & (o[k] < hi[k])
        assert ok.sum() == 0 or (off[ok].min() >= 0 and off[ok].max() < flat.size), "descriptor leaves the volume"
        v = np.zeros(dims, np.float32)
        v[ok] = flat[off[ok]].astype(np.float32)
        out[i] = v * np.float32(scale) + np.float32(bias)
    return torch.from_numpy(out)


def emu_axis_resample(x, axis, idx, w, validated=False):
    a = np.moveaxis(x.detach().cpu().numpy(), axis, 0)
    idx, w = idx.cpu().numpy(), w.cpu().numpy()
    out = np.zeros((idx.shape[0],) + a.shape[1:], np.float32)
    for j in range(idx.shape[0]):
        acc = np.zeros(a.shape[1:], np.float32)
        for t in range(idx.shape[1]):
            if idx[j, t] >= 0:
                acc = acc + w[j, t] * a[idx[j, t]]
        out[j] = acc
    return torch.from_numpy(np.ascontiguousarray(np.moveaxis(out, 0, axis)))
