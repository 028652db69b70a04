"""Two iterations of the reference's `train_sr` loop body (train_all.py:118-139), written with the reference's
own import lines (train_all.py:20-21,29), run in a fresh interpreter that only has rehrseg_amd/ on sys.path --
the drop-in of INTEGRATION.md section 1 executing on the HIP kernels.  The first loss is checked against the
CPU oracle's composition of the same expression."""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LOOP = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, {pkg!r})
    import torch
    from models.FLAVR.FLAVR_arch import UNet_3D_3D
    from models.seg_model import SegModel, Distiller
    from utils.seg_utils import zscore_normalization, BCEDiceLoss, _build_loss
    device = torch.device("cuda:0")
    blob = torch.load({blob!r})
    model = UNet_3D_3D(2, "unet_18", 4, 4, batchnorm=False, joinType="concat", upmode="transpose",
                       use_uncertainty=False)
    model.load_state_dict(blob["sd"])
    model = model.to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=10)
    loss_obj, loss_seg = torch.nn.L1Loss(), BCEDiceLoss(1.0, 1.0)
    slice_separation, num_slices, losses = 4.0, 4, []
    for i, (patches_lr, patches_hr) in enumerate([(blob["lr"], blob["hr"])] * 2):
        patches_hr = patches_hr.to(device)
        patches_lr = patches_lr.clone().to(device)
        patches_hr = patches_hr[:, :, int(slice_separation) * (num_slices // 2 - 1):int(slice_separation) * (num_slices // 2), ...]
        patches_hr_hat = model(patches_lr)
        loss = loss_obj(patches_hr_hat[:, 0:1, ...], patches_hr[:, 0:1, ...])
        loss += loss_seg(patches_hr_hat[:, 1:, ...], patches_hr[:, 1:, ...]) * 1.0
        opt.zero_grad()
        loss.backward()
        opt.step()
        scheduler.step()
        losses.append(float(loss))
    from rehrseg_amd import hip_backend
    print("RESULT " + json.dumps({{"losses": losses, "wino": hip_backend.wino_launches}}))
""")


def test_train_sr_loop_body_with_reference_imports(tmp_path):
    from oracle import aux_oracle as ao
    from oracle import flavr_oracle as fo
    from oracle.detinit import det_input, det_tensor
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    m = UNet_3D_3D(2, "unet_18", 4, 4)
    sd = {k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()}
    lr_p = det_input("dropin.lr", (2, 2, 4, 32, 32), "rand")
    hr_p = det_input("dropin.hr", (2, 2, 16, 32, 32), "rand")
    hr_p[:, 1:] = (hr_p[:, 1:] > 0.5).float()
    blob = str(tmp_path / "blob.pt")
    torch.save({"sd": sd, "lr": lr_p, "hr": hr_p}, blob)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", LOOP.format(pkg=os.path.join(ROOT, "rehrseg_amd"), blob=blob)],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    # oracle composition of train_all.py:122-134 for the first iteration
    hat = fo.unet_3d_3d(sd, lr_p.clone(), 2, 4, 4, False)
    hr = hr_p[:, :, 4:8]
    ref = (hat[:, 0:1] - hr[:, 0:1]).abs().mean() + ao.bce_dice(hat[:, 1:], hr[:, 1:])
    assert abs(res["losses"][0] - ref.item()) <= 1e-4 * abs(ref.item()), (res, ref.item())
    assert res["losses"][1] != res["losses"][0] and all(l == l for l in res["losses"])  # the step moved the weights
    assert res["wino"] > 0  # the HIP kernels ran (there is no other device path)
