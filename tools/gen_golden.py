"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference's Python is imported from /root/reference with inert placeholder
modules for its absent third-party imports; it is loaded
with the deterministic name-keyed parameters of oracle/detinit.py and run on
deterministic inputs.  Only inputs/outputs/gradient summaries are written --
never any reference source.  The GPU box has no /root/reference: tests there
read the committed fixtures.

    python tools/gen_golden.py            # rewrites tests/golden/
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.detinit import det_input, det_tensor  # noqa: E402

sys.dont_write_bytecode = True
ABSENT = ("resize", "SimpleITK", "nibabel", "h5py", "omegaconf", "degrade", "acvl_utils", "nnunetv2",
          "batchgenerators", "batchgeneratorsv2", "dynamic_network_architectures", "kornia", "torchvision")


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError("absent third-party symbol called")


class _Mod(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (_Dummy,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)

    def create_module(self, spec):
        return _Mod(spec.name)

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference")
    import models.FLAVR.FLAVR_arch as fa
    return fa


OUT = os.path.join(ROOT, "tests", "golden")
FULL_GRADS = ("encoder.stem.0.bias", "outconv.1.weight", "decoder.4.upconv.1.attn_layer.0.bias",
              "encoder.layer2.0.downsample.0.weight", "decoder.1.upconv.0.bias", "uncertainty_out.weight",
              "feature_fuse1.conv.0.bias")


def load_det(model):
    sd = {k: det_tensor(k, tuple(v.shape)) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)


def flavr_case(fa, tag, img_channels, n_inputs, n_outputs, unc, hw):
    torch.manual_seed(0)
    m = fa.UNet_3D_3D(img_channels, "unet_18", n_inputs, n_outputs, use_uncertainty=unc)
    load_det(m)
    x = det_input(tag + ".x", (1, img_channels, n_inputs, hw, hw), "rand")
    tgt = det_input(tag + ".t", (1, img_channels, n_outputs, hw, hw), "rand")
    xin = x.clone()
    out = m(xin)
    rec = {"x": x.numpy(), "x_after": xin.numpy(), "target": tgt.numpy()}
    if unc:
        out, sigma = out
        loss = (out - tgt).abs().mean() + sigma.mean()
        rec["sigma"] = sigma.detach().numpy()
    else:
        loss = (out - tgt).abs().mean()
    rec["out"] = out.detach().numpy()
    rec["loss"] = np.float64(loss.item())
    loss.backward()
    names, norms = [], []
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if k in FULL_GRADS:
            rec["grad:" + k] = p.grad.numpy()
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms, dtype=np.float64)
    feats = m(x.clone(), return_inetermediate_feature=True)
    for i, f in enumerate(feats):
        f = f.detach()
        rec[f"feat{i}_mean"] = f.double().mean((2, 3, 4)).numpy()
        rec[f"feat{i}_slice"] = f[0, :8, 1, :8, :8].numpy()
    rec["meta"] = np.array([img_channels, n_inputs, n_outputs, int(unc), hw])
    np.savez_compressed(os.path.join(OUT, f"flavr_{tag}.npz"), **rec)
    print(tag, "loss", loss.item(), "params with grad", len(names))


def main():
    os.makedirs(OUT, exist_ok=True)
    fa = import_reference()
    flavr_case(fa, "c2_n4", 2, 4, 4, False, 32)
    flavr_case(fa, "c2_n4_unc", 2, 4, 4, True, 32)
    flavr_case(fa, "c1_n8", 1, 8, 4, False, 32)


if __name__ == "__main__":
    main()
