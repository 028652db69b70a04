"""G8: the reference's own SegModel.forward / MyUnetDecoder.forward /
DC_and_weighted_CE_loss.forward, captured by tools/gen_golden_segmodel.py over eager-torch stand-ins for
the absent third-party bases.  Checked here: the CPU oracle, and the product's host wiring through the
test-only ABI emulation.  (`-m gpu` twin: tests/test_segmodel_golden_gpu.py.)
Tolerances: forward 1e-5 of the tensor's max (fp32 CPU vs fp32 CPU), loss 1e-5, gradient norms 1e-3 --
the reference's own fp32 run is the yardstick, so its InstanceNorm-stack rounding noise (DESIGN.md
section 5) is part of the fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import aux_oracle as ao
from oracle import segmodel_oracle as so
from oracle.detinit import det_input, det_state_dict
from rehrseg_amd.utils import seg_utils as su
from test_segmodel_cpu import build, canonical

GDIR = os.path.join(os.path.dirname(__file__), "golden")
SMALL = dict(n_stages=3, features_per_stage=[32, 64, 96], kernel_sizes=[[1, 3, 3], [3, 3, 3], [3, 3, 3]],
             strides=[[1, 1, 1], [1, 2, 2], [2, 2, 2]], n_conv_per_stage=[2, 2, 2], n_conv_per_stage_decoder=[2, 2],
             num_classes=2, upscale=4)
ANISO4 = dict(n_stages=4, features_per_stage=[32, 64, 128, 160],
              kernel_sizes=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3]],
              strides=[[1, 1, 1], [1, 2, 2], [1, 2, 2], [2, 2, 2]], n_conv_per_stage=[2, 2, 2, 2],
              n_conv_per_stage_decoder=[2, 2, 2], num_classes=2, upscale=4)
CASES = {"small": SMALL, "aniso4": ANISO4}


def relmax(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def fixture(tag):
    return np.load(os.path.join(GDIR, f"segmodel_{tag}.npz"))


def stage2_loss(out, out_up, skip1, G, tag, loss_fn, device="cpu"):
    """The fixture's loss: DC + uncertainty-CE on the LR head, DC + CE on the HR head (train_all.py:538-548)
    plus a fixed random projection of skips[1] (the tensor the Distiller consumes)."""
    t = lambda k: torch.from_numpy(G[k]).to(device)
    g3 = det_input(tag + ".g3", tuple(skip1.shape), "randn").to(device)
    l_lr = loss_fn(out, t("lab_lr"), t("unc"))
    l_hr = loss_fn(out_up, t("lab_hr"), None)
    return l_lr + l_hr + (skip1 * g3).mean(), l_lr, l_hr


def check_against_fixture(tag, out, out_up, skips, losses, grads, fwd_tol, grad_tol):
    G = fixture(tag)
    assert relmax(out.detach().cpu(), G["out"]) < fwd_tol and relmax(out_up.detach().cpu(), G["out_up"]) < fwd_tol
    assert relmax(skips[1].detach().cpu(), G["skip1"]) < fwd_tol
    for i, s in enumerate(skips):
        assert tuple(s.shape) == tuple(G[f"skip{i}_shape"])
        assert relmax(s.detach().double().mean((2, 3, 4)).cpu(), G[f"skip{i}_mean"]) < 10 * fwd_tol
    for v, k in zip(losses, ("loss", "loss_lr", "loss_hr")):
        v = float(v.detach())
        assert abs(v - float(G[k])) <= 1e-5 * max(1.0, abs(float(G[k]))), (k, v, float(G[k]))
    worst = ("", 0.0)
    for name, ref in zip(G["grad_names"], G["grad_norms"]):
        name = str(name)
        if "conv.bias" in name and "sr_head" not in name:   # identically 0 behind InstanceNorm: rounding noise only
            continue
        got = float(grads[name].double().norm())
        assert abs(got - ref) <= grad_tol * max(ref, 1e-12), (name, got, float(ref))
        full = "grad:" + name
        if full in G.files:
            e = float((grads[name].double().cpu() - torch.from_numpy(G[full]).double()).norm()) / max(ref, 1e-12)
            worst = max(worst, (name, e), key=lambda t: t[1])
            assert e <= grad_tol, (name, e)
    return worst


@pytest.mark.parametrize("tag", ["small", "aniso4"])
def test_oracle_against_reference_segmodel(tag):
    cfg, G = CASES[tag], fixture(tag)
    sd = {k: v.requires_grad_() for k, v in det_state_dict(so.segmodel_shapes(cfg)).items()}
    out, out_up, skips = so.seg_model(sd, torch.from_numpy(G["x"]), cfg, return_features=True)
    loss, l_lr, l_hr = stage2_loss(out, out_up, skips[1], G, tag, ao.dc_and_weighted_ce)
    loss.backward()
    check_against_fixture(tag, out, out_up, skips, (loss, l_lr, l_hr), {k: v.grad for k, v in sd.items()}, 1e-5, 1e-3)
    # deep supervision: finest first, every level through its own seg layer (ref seg_model.py:40-51)
    with torch.no_grad():
        outs, _ = so.seg_model(sd, torch.from_numpy(G["x"]), dict(cfg, deep_supervision=True))
    for i, o in enumerate(outs):
        assert relmax(o, G[f"ds_out{i}"]) < 1e-5
    assert float(G["ds_out_up_maxdiff"]) == 0.0


@pytest.mark.parametrize("tag", ["small", "aniso4"])
def test_product_host_wiring_against_reference_segmodel(tag, emu):
    cfg, G = CASES[tag], fixture(tag)
    m, _ = build(cfg)
    out, out_up, skips = m(torch.from_numpy(G["x"]).clone(), return_inetermediate_feature=True)
    loss, l_lr, l_hr = stage2_loss(out, out_up, skips[1], G, tag, su._build_loss())
    loss.backward()
    grads = {canonical(k): p.grad for k, p in m.named_parameters()}
    check_against_fixture(tag, out, out_up, skips, (loss, l_lr, l_hr), grads, 1e-5, 1e-3)
    mds, _ = build(cfg, deep_supervision=True)
    with torch.no_grad():
        outs, _ = mds(torch.from_numpy(G["x"]).clone())
    for i, o in enumerate(outs):
        assert relmax(o, G[f"ds_out{i}"]) < 1e-5


def loss_cases():
    L = np.load(os.path.join(GDIR, "seg_losses.npz"))
    for C in (2, 3):
        for u in (0, 1):
            for wd in (1.0, 0.5):
                tag = f"c{C}_u{u}_w{int(wd * 10)}"
                yield tag, wd, L[tag + ".logits"], L[tag + ".target"], (L[tag + ".unc"] if u else None), \
                    float(L[tag + ".loss"]), L[tag + ".grad"]


def test_dc_ce_composition_oracle_and_product_against_reference():
    for tag, wd, lg, tg, un, ref, rgrad in loss_cases():
        for fn in (lambda a, b, c: ao.dc_and_weighted_ce(a, b, c, 1.0, wd), su._build_loss(weight_dice=wd)):
            x = torch.from_numpy(lg).requires_grad_()
            v = fn(x, torch.from_numpy(tg), None if un is None else torch.from_numpy(un))
            v.backward()
            assert abs(float(v) - ref) <= 1e-6 * max(1.0, abs(ref)), (tag, float(v), ref)
            assert relmax(x.grad, rgrad) < 1e-5, tag
