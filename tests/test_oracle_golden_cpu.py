"""The CPU oracle (oracle/flavr_oracle.py) against the fixtures that
tools/gen_golden.py captured from the reference itself."""
import os

import numpy as np
import pytest
import torch

from oracle import flavr_oracle as fo
from oracle.detinit import det_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["c2_n4", "c2_n4_unc", "c1_n8"]


def run_oracle(g):
    ic, ni, no, unc, hw = (int(v) for v in g["meta"])
    shapes = fo.flavr_shapes(ic, ni, no, bool(unc))
    sd = {k: v.requires_grad_() for k, v in det_state_dict(shapes).items()}
    x = torch.from_numpy(g["x"]).clone()
    out = fo.unet_3d_3d(sd, x, ic, ni, no, bool(unc))
    tgt = torch.from_numpy(g["target"])
    if unc:
        out, sigma = out
        loss = (out - tgt).abs().mean() + sigma.mean()
    else:
        sigma = None
        loss = (out - tgt).abs().mean()
    loss.backward()
    return sd, x, out, sigma, loss


@pytest.mark.parametrize("tag", CASES)
def test_flavr_oracle_matches_reference(tag):
    g = np.load(os.path.join(GOLD, f"flavr_{tag}.npz"))
    sd, x, out, sigma, loss = run_oracle(g)
    assert np.allclose(x.numpy(), g["x_after"], atol=1e-6)  # in-place mean subtraction of channel 0
    assert np.allclose(out.detach().numpy(), g["out"], rtol=1e-5, atol=1e-6)
    if sigma is not None:
        assert np.allclose(sigma.detach().numpy(), g["sigma"], rtol=1e-5, atol=1e-6)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    for k, n in norms.items():
        got = float(sd[k].grad.double().norm())
        assert abs(got - n) <= 1e-4 * max(n, 1e-6) + 1e-9, (k, got, n)
    for key in g.files:
        if key.startswith("grad:"):
            assert np.allclose(sd[key[5:]].grad.numpy(), g[key], rtol=1e-4, atol=1e-7), key


@pytest.mark.parametrize("tag", CASES)
def test_flavr_oracle_encoder_features(tag):
    g = np.load(os.path.join(GOLD, f"flavr_{tag}.npz"))
    ic, ni, no, unc, hw = (int(v) for v in g["meta"])
    sd = det_state_dict(fo.flavr_shapes(ic, ni, no, bool(unc)))
    with torch.no_grad():
        feats = fo.unet_3d_3d(sd, torch.from_numpy(g["x"]).clone(), ic, ni, no, bool(unc),
                              return_intermediate_feature=True)
    for i, f in enumerate(feats):
        assert np.allclose(f.double().mean((2, 3, 4)).numpy(), g[f"feat{i}_mean"], rtol=1e-5, atol=1e-7)
        assert np.allclose(f[0, :8, 1, :8, :8].numpy(), g[f"feat{i}_slice"], rtol=1e-5, atol=1e-6)


def test_state_dict_key_set_matches_reference():
    g = np.load(os.path.join(GOLD, "flavr_c2_n4_unc.npz"))
    have = set(fo.flavr_shapes(2, 4, 4, True))
    # every parameter that received a gradient in the reference exists in the oracle's layout
    assert set(g["grad_names"].tolist()) <= have
