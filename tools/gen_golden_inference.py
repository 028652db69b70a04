"""Fixtures for the whole-volume inference helpers, produced by running the
REFERENCE's own functions in the build container (import recipe: tools/gen_golden.py):

  utils/sr_utils.py:102-135   apply_to_vol_flavr         (with the real UNet_3D_3D, deterministic weights)
  utils/seg_utils.py:176-199  compute_steps_for_sliding_window
  utils/seg_utils.py:201-227  _internal_maybe_mirror_and_predict
  utils/seg_utils.py:229-238  _internal_get_sliding_window_slicers
  utils/seg_utils.py:240-287  _internal_predict_sliding_window_return_logits (use_gaussian=False: the Gaussian
                              comes from nnunetv2, absent offline -> that branch stays unpinned)

apply_to_vol_flavr hard-codes `.cuda()` on its zero pads; there is no GPU in the build container, so
torch.Tensor.cuda is made the identity for the duration of that one call (nothing else is touched).

    python tools/gen_golden_inference.py     # rewrites tests/golden/inference_paths.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gen_golden import OUT, import_reference, load_det  # noqa: E402
from oracle.detinit import det_input  # noqa: E402
from toy_models import ToySegNet  # noqa: E402


def main():
    fa = import_reference()
    import utils.seg_utils as su
    import utils.sr_utils as sr

    rec = {}
    # ---- apply_to_vol_flavr: volume (slices, C, X, Y) = (6, 2, 20, 18) -> padded to 32 x 32 internally
    model = fa.UNet_3D_3D(2, "unet_18", 4, 4, batchnorm=False, joinType="concat", upmode="transpose",
                          use_uncertainty=True).eval()
    load_det(model)
    vol = det_input("vol.x", (6, 2, 20, 18), "rand")
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        out0 = sr.apply_to_vol_flavr(model, vol.clone(), 0)
        out1 = sr.apply_to_vol_flavr(model, vol.clone(), 1)
    finally:
        torch.Tensor.cuda = orig_cuda
    rec.update(vol_in=vol.numpy(), vol_out0=out0.numpy(), vol_out1=out1.numpy())

    # ---- sliding-window geometry
    cases = [((20, 45, 63), (14, 32, 48), 0.5), ((14, 320, 384), (14, 320, 384), 0.5), ((9, 70, 33), (8, 32, 32), 0.75)]
    for i, (img, tile, step) in enumerate(cases):
        steps = su.compute_steps_for_sliding_window(img, tile, step)
        for a in range(3):
            rec[f"steps{i}_{a}"] = np.asarray(steps[a], dtype=np.int64)
        sl = su._internal_get_sliding_window_slicers(img, patch_size=list(tile), tile_step_size=step)
        rec[f"slicers{i}"] = np.asarray([[s.start for s in t[1:]] + [s.stop for s in t[1:]] for t in sl], dtype=np.int64)

    # ---- mirror TTA + tiled predictor with a toy network (LR head: out_idx 0, HR head: out_idx 1, sep 2)
    net = ToySegNet(sep=2)
    x = det_input("tta.x", (1, 1, 6, 12, 10))
    rec["tta_in"] = x.numpy()
    rec["tta_lr"] = su._internal_maybe_mirror_and_predict(net, x.clone(), 0, deep_supervision=False).numpy()
    rec["tta_hr"] = su._internal_maybe_mirror_and_predict(net, x.clone(), 1, deep_supervision=False).numpy()
    data = det_input("tile.x", (1, 10, 21, 19))
    rec["tile_in"] = data.numpy()
    patch = [6, 12, 10]
    sl = su._internal_get_sliding_window_slicers(data.shape[1:], patch_size=patch)
    rec["tile_lr"] = su._internal_predict_sliding_window_return_logits(
        data.clone(), sl, net, False, 0, 1, patch, use_gaussian=False, deep_supervision=False).float().numpy()
    sl = su._internal_get_sliding_window_slicers(data.shape[1:], patch_size=patch)
    rec["tile_hr"] = su._internal_predict_sliding_window_return_logits(
        data.clone(), sl, net, False, 1, 2, [patch[0] * 2, patch[1], patch[2]]).float().numpy()

    path = os.path.join(OUT, "inference_paths.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, {k: v.shape for k, v in rec.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
