"""SegModel host wiring (virtual concat into the decoder stages, transposed convs with
kernel = stride, thin-output heads, depth upsample) with the C-ABI emulated on the CPU,
against the functional oracle (parity of the nnU-Net bases is UNPINNED, see the oracle)."""
import pytest
import torch
import torch.nn as nn

from oracle import segmodel_oracle as so
from oracle.detinit import det_input, det_tensor
from rehrseg_amd.models.seg_model import Distiller, SegModel

SMALL = dict(n_stages=3, features_per_stage=[32, 64, 96], kernel_sizes=[[1, 3, 3], [3, 3, 3], [3, 3, 3]],
             strides=[[1, 1, 1], [1, 2, 2], [2, 2, 2]], n_conv_per_stage=[2, 2, 2], n_conv_per_stage_decoder=[2, 2],
             num_classes=2, upscale=4)


def canonical(key):
    """Primary name of a (possibly aliased) nnU-Net state-dict key."""
    if key.startswith("decoder.encoder."):
        key = key[len("decoder."):]
    return key.replace("all_modules.0.", "conv.").replace("all_modules.1.", "norm.")


def build(cfg, device="cpu", deep_supervision=False):
    m = SegModel(input_channels=1, num_classes=cfg["num_classes"], n_stages=cfg["n_stages"], upscale=cfg["upscale"],
                 features_per_stage=cfg["features_per_stage"], conv_op=nn.Conv3d, kernel_sizes=cfg["kernel_sizes"],
                 strides=cfg["strides"], n_conv_per_stage=cfg["n_conv_per_stage"],
                 n_conv_per_stage_decoder=cfg["n_conv_per_stage_decoder"], conv_bias=True, norm_op=nn.InstanceNorm3d,
                 norm_op_kwargs={"eps": 1e-5, "affine": True}, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=nn.LeakyReLU, nonlin_kwargs={"inplace": True}, deep_supervision=deep_supervision)
    sd = {k: det_tensor(canonical(k), tuple(v.shape)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    return m.to(device), sd


def check(cfg, shape, device, tol):
    m, sd = build(cfg, device)
    x = det_input("seg.x", shape, "randn")
    out, out_up, skips = m(x.clone().to(device), return_inetermediate_feature=True)
    # random projections as the loss: a plain mean of InstanceNorm-ed features makes the gradients
    # cancel to ~0 and turns any fp32 rounding into a huge relative error (also in torch's own fp32)
    g1, g2, g3 = (det_input(n, tuple(t.shape), "randn") for n, t in (("g1", out), ("g2", out_up), ("g3", skips[1])))
    loss = (out * g1.to(device)).mean() + (out_up * g2.to(device)).mean() + (skips[1] * g3.to(device)).mean()
    loss.backward()
    # Two oracles: fp64 is the reference value; fp32 (the reference's own CPU precision) calibrates how
    # much of the deviation is plain fp32 rounding.  Deep InstanceNorm stacks amplify rounding in the
    # backward pass: torch's fp32 gradients themselves sit ~4e-3 from fp64 on the 6-stage plans, so the
    # gradient bar is "within 1e-3, or within 6x the fp32 oracle's own distance from fp64" (sums that
    # cancel, e.g. a transposed-conv bias in front of an InstanceNorm, are the worst case: 4x observed).
    runs = {}
    for dt in (torch.float64, torch.float32):
        osd = {k: v.to(dt).requires_grad_() for k, v in sd.items() if k in so.segmodel_shapes(cfg)}
        r = so.seg_model(osd, x.to(dt), cfg, return_features=True)
        ((r[0] * g1.to(dt)).mean() + (r[1] * g2.to(dt)).mean() + (r[2][1] * g3.to(dt)).mean()).backward()
        runs[dt] = (osd, r)
    osd, (r_out, r_up, r_skips) = runs[torch.float64]
    osd32 = runs[torch.float32][0]

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))
    assert out.shape == r_out.shape and out_up.shape == r_up.shape
    assert rel(out, r_out) < tol and rel(out_up, r_up) < tol
    for a, b in zip(skips, r_skips):
        assert rel(a, b) < tol
    params = dict(m.named_parameters())
    for k, v in osd.items():
        if v.grad is None or "conv.bias" in k:   # d(conv bias) is identically 0 behind InstanceNorm
            continue
        n = float(v.grad.norm())
        err = float((params[k].grad.cpu().double() - v.grad).norm())
        err32 = float((osd32[k].grad.double() - v.grad).norm())
        assert err <= max(10 * tol * n, 6 * err32) + 1e-7, (k, err / n, err32 / n)


def test_segmodel_small_plan(emu):
    check(SMALL, (2, 1, 4, 16, 16), "cpu", 1e-4)


def test_state_dict_has_nnunet_keys_and_aliases():
    m, sd = build(SMALL)
    keys = set(sd)
    for k in so.segmodel_shapes(SMALL):
        assert k in keys, k
    assert "encoder.stages.0.0.convs.0.all_modules.0.weight" in keys      # Sequential alias of conv
    assert "encoder.stages.0.0.convs.0.all_modules.1.bias" in keys        # ... and of norm
    assert "decoder.encoder.stages.1.0.convs.1.conv.weight" in keys       # decoder keeps the encoder
    assert m.state_dict()["decoder.encoder.stages.0.0.convs.0.conv.weight"].data_ptr() == \
        m.encoder.stages[0][0].convs[0].conv.weight.data_ptr()
    # nnU-Net checkpoints are loaded with strict=False (train_all.py:499): primary keys alone must load
    prim = {k: v for k, v in sd.items() if "all_modules" not in k and not k.startswith("decoder.encoder")}
    missing, unexpected = m.load_state_dict(prim, strict=False)
    assert not unexpected


def test_deep_supervision_outputs(emu):
    m, _ = build(SMALL, deep_supervision=True)
    out, out_up = m(det_input("ds.x", (1, 1, 4, 16, 16)))
    assert isinstance(out, list) and [tuple(o.shape) for o in out] == [(1, 2, 4, 16, 16), (1, 2, 4, 8, 8)]
    assert tuple(out_up.shape) == (1, 2, 16, 16, 16)
    m.decoder.deep_supervision = False   # train_all.py:562 toggles it for evaluation
    out, _ = m(det_input("ds.x", (1, 1, 4, 16, 16)))
    assert tuple(out.shape) == (1, 2, 4, 16, 16)
