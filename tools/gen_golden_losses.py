"""Fixtures for the rows next to the conv path: Distiller, the in-reference
losses, zscore_normalization and the teacher pass get_intermediate_features -- produced by
running the REFERENCE (build container only; see tools/gen_golden.py for the import recipe).

    python tools/gen_golden_losses.py        # rewrites tests/golden/aux_*.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden import OUT, import_reference, load_det  # noqa: E402
from oracle.detinit import det_input, det_tensor  # noqa: E402


def main():
    fa = import_reference()
    import models.seg_model as sm
    import utils.seg_utils as su

    rec = {}
    # --- Distiller (models/seg_model.py:115-151) with configs/brain.yaml's lambdas
    torch.manual_seed(0)
    dist = sm.Distiller(64, 64, lambda_l1=0.0, lambda_cosine=1.0, lambda_structure=1.0)
    load_det(dist)
    fs = det_input("dist.s", (2, 64, 6, 16, 16)).requires_grad_()
    ft = det_input("dist.t", (2, 64, 6, 16, 16))
    loss = dist(fs, ft)
    loss.backward()
    rec.update(dist_loss=np.float64(loss.item()), dist_grad_fs=fs.grad.numpy(),
               dist_grad_w=dist.distill.weight.grad.numpy(), dist_grad_b=dist.distill.bias.grad.numpy())
    dist2 = sm.Distiller(64, 64, lambda_l1=0.5, lambda_cosine=0.0, lambda_structure=0.0)
    load_det(dist2)
    rec["dist_l1_loss"] = np.float64(dist2(fs.detach(), ft).item())

    # --- BCEDiceLoss (utils/seg_utils.py:786-885), used by train_sr on the seg channel
    bd = su.BCEDiceLoss(1.0, 1.0)
    logits = det_input("bcedice.x", (2, 1, 4, 16, 16)).requires_grad_()
    target = det_input("bcedice.t", (2, 1, 4, 16, 16), "randint2")
    l = bd(logits, target)
    l.backward()
    rec.update(bcedice_loss=np.float64(l.item()), bcedice_grad=logits.grad.numpy())

    # --- RobustCrossEntropyLoss with the (B,B,...) uncertainty broadcast (seg_utils.py:289-304, :349)
    ce = su.RobustCrossEntropyLoss(reduction="none")
    lg = det_input("ce.x", (2, 2, 4, 8, 8)).requires_grad_()
    tg = det_input("ce.t", (2, 1, 4, 8, 8), "randint2")
    un = det_input("ce.u", (2, 1, 4, 8, 8), "rand")
    l1 = ce(lg, tg[:, 0], un)
    l1.backward()
    rec.update(ce_unc_loss=np.float64(l1.item()), ce_unc_grad=lg.grad.numpy(),
               ce_plain_loss=np.float64(ce(lg.detach(), tg[:, 0], None).item()))

    # --- zscore_normalization (seg_utils.py:137-156): in place on the caller's tensor
    z = det_input("z.x", (2, 1, 6, 16, 16), "rand") * 3 + 1
    zin = z.clone()
    zo = su.zscore_normalization(zin)
    rec.update(zscore_out=zo.numpy(), zscore_in_after=zin.numpy(), zscore_in=z.numpy())

    # --- get_intermediate_features (train_all.py:85-112) on a small teacher
    import train_all as ta
    torch.manual_seed(0)
    teacher = fa.UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True).eval()
    load_det(teacher)
    img = det_input("gif.img", (2, 1, 6, 32, 32), "rand")
    lab = det_input("gif.lab", (2, 1, 6, 32, 32), "randint2")
    img_in = img.clone()
    with torch.no_grad():
        feats = ta.get_intermediate_features(teacher, img_in, lab, torch.device("cpu"))
    rec["gif_img_after"] = img_in.numpy()
    for i in feats:
        f = feats[i]
        rec[f"gif_shape{i}"] = np.array(f.shape)
        rec[f"gif_mean{i}"] = f.double().mean((3, 4)).numpy()
    rec["gif_feat1"] = feats[1].numpy()  # the only one the stage-2 loop consumes (train_all.py:550)
    np.savez_compressed(os.path.join(OUT, "aux_losses_teacher.npz"), **rec)
    print({k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in rec.items()})


if __name__ == "__main__":
    main()
