"""Host logic of rehrseg_amd.ops (phases, taps, packing, virtual concat, fused
tails) against torch.nn.functional, with the C-ABI emulated on the CPU in fp64."""
import pytest
import torch
import torch.nn.functional as F

from oracle.flavr_oracle import uasr_head
from rehrseg_amd import ops

torch.manual_seed(0)
DT = torch.float64


def _rand(*s):
    return torch.randn(*s, dtype=DT)


def _grads(out, inputs):
    g = torch.randn_like(out)
    return torch.autograd.grad(out, inputs, g, allow_unused=True), g


def _cmp(a, b, tol=1e-9):
    assert a.shape == b.shape
    assert torch.allclose(a, b, rtol=tol, atol=tol), float((a - b).abs().max())


CONVS = [
    # Cin, Cout, K, stride, pad, dims
    (32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (3, 6, 7)),
    (64, 32, (3, 3, 3), (1, 2, 2), (1, 1, 1), (4, 9, 8)),      # layer2/3 first conv
    (32, 64, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 8, 7)),      # downsample projection
    (32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 5, 5)),      # anisotropic nnU-Net kernel
    (32, 96, (3, 3, 3), (2, 2, 2), (1, 1, 1), (5, 6, 7)),      # nnU-Net strided stage, odd extents
    (32, 32, (5, 5, 5), (1, 1, 1), (2, 2, 2), (4, 5, 6)),
]


@pytest.mark.parametrize("Cin,Cout,K,stride,pad,dims", CONVS)
def test_conv3d_fwd_bwd(emu, Cin, Cout, K, stride, pad, dims):
    x = _rand(2, Cin, *dims).requires_grad_()
    w = _rand(Cout, Cin, *K).requires_grad_()
    b = _rand(Cout).requires_grad_()
    y = ops.fused_conv3d(x, w, b, stride, pad, act=ops.ACT_LRELU, slope=0.2)
    ref = F.leaky_relu(F.conv3d(x, w, b, stride, pad), 0.2)
    _cmp(y, ref)
    g = torch.randn_like(ref)
    got = torch.autograd.grad(y, (x, w, b), g)
    exp = torch.autograd.grad(ref, (x, w, b), g)
    for a, e in zip(got, exp):
        _cmp(a, e)


TCONVS = [
    (64, 32, (3, 4, 4), (1, 2, 2), (1, 1, 1), (3, 5, 6)),   # FLAVR upConv3D
    (64, 32, (2, 2, 2), (2, 2, 2), (0, 0, 0), (3, 4, 5)),   # nnU-Net transpconv kernel = stride
    (32, 32, (1, 2, 2), (1, 2, 2), (0, 0, 0), (2, 3, 3)),
]


@pytest.mark.parametrize("Cin,Cout,K,stride,pad,dims", TCONVS)
def test_conv_transpose3d_fwd_bwd(emu, Cin, Cout, K, stride, pad, dims):
    x = _rand(2, Cin, *dims).requires_grad_()
    w = _rand(Cin, Cout, *K).requires_grad_()
    b = _rand(Cout).requires_grad_()
    y = ops.fused_conv3d(x, w, b, stride, pad, transposed=True)
    ref = F.conv_transpose3d(x, w, b, stride, pad)
    _cmp(y, ref)
    g = torch.randn_like(ref)
    got = torch.autograd.grad(y, (x, w, b), g)
    exp = torch.autograd.grad(ref, (x, w, b), g)
    for a, e in zip(got, exp):
        _cmp(a, e)


@pytest.mark.parametrize("transposed", [False, True])
def test_virtual_concat(emu, transposed):
    x1 = _rand(1, 32, 3, 4, 5).requires_grad_()
    x2 = _rand(1, 64, 3, 4, 5).requires_grad_()
    if transposed:
        w = _rand(96, 32, 3, 4, 4).requires_grad_()
        args = ((1, 2, 2), (1, 1, 1))
        ref = F.conv_transpose3d(torch.cat([x1, x2], 1), w, None, *args)
    else:
        w = _rand(64, 96, 3, 3, 3).requires_grad_()
        args = ((1, 1, 1), (1, 1, 1))
        ref = F.conv3d(torch.cat([x1, x2], 1), w, None, *args)
    y = ops.fused_conv3d(x1, w, None, *args, x2=x2, transposed=transposed)
    _cmp(y, ref)
    g = torch.randn_like(ref)
    got = torch.autograd.grad(y, (x1, x2, w), g)
    exp = torch.autograd.grad(ref, (x1, x2, w), g)
    for a, e in zip(got, exp):
        _cmp(a, e)


def _se_ref(v, aw, ab):
    m = v.mean((2, 3, 4), keepdim=True)
    return v * torch.sigmoid(F.conv3d(m, aw, ab))


@pytest.mark.parametrize("with_res,act", [(True, ops.ACT_RELU), (False, ops.ACT_LRELU)])
def test_conv_se_block(emu, with_res, act):
    x = _rand(2, 32, 3, 5, 4).requires_grad_()
    w = _rand(32, 32, 3, 3, 3).requires_grad_()
    b = _rand(32).requires_grad_()
    aw = _rand(32, 32, 1, 1, 1).requires_grad_()
    ab = _rand(32).requires_grad_()
    res = _rand(2, 32, 3, 5, 4).requires_grad_() if with_res else None
    y = ops.fused_conv3d(x, w, b, 1, 1, se=(aw, ab), res=res, act=act, slope=0.2)
    v = _se_ref(F.conv3d(x, w, b, 1, 1), aw, ab)
    if with_res:
        v = v + res
    ref = torch.relu(v) if act == ops.ACT_RELU else F.leaky_relu(v, 0.2)
    _cmp(y, ref)
    ins = [x, w, b, aw, ab] + ([res] if with_res else [])
    g = torch.randn_like(ref)
    got = torch.autograd.grad(y, ins, g)
    exp = torch.autograd.grad(ref, ins, g)
    for a, e in zip(got, exp):
        _cmp(a, e, 1e-8)


def test_conv_instnorm_lrelu_block(emu):
    x = _rand(2, 32, 4, 5, 6).requires_grad_()
    w = _rand(64, 32, 3, 3, 3).requires_grad_()
    b = _rand(64).requires_grad_()
    ga = _rand(64).requires_grad_()
    be = _rand(64).requires_grad_()
    y = ops.fused_conv3d(x, w, b, (1, 2, 2), 1, inorm=(ga, be), eps=1e-5, act=ops.ACT_LRELU, slope=0.01)
    ref = F.leaky_relu(F.instance_norm(F.conv3d(x, w, b, (1, 2, 2), 1), weight=ga, bias=be, eps=1e-5), 0.01)
    _cmp(y, ref, 1e-8)
    g = torch.randn_like(ref)
    got = torch.autograd.grad(y, (x, w, b, ga, be), g)
    exp = torch.autograd.grad(ref, (x, w, b, ga, be), g)
    for a, e in zip(got, exp):
        _cmp(a, e, 1e-7)


def test_thin_input_conv(emu):
    x = _rand(2, 1, 4, 10, 9)
    w = _rand(64, 1, 3, 7, 7).requires_grad_()
    b = _rand(64).requires_grad_()
    y = ops.fused_conv3d(x, w, b, (1, 2, 2), (1, 3, 3), act=ops.ACT_RELU)
    ref = torch.relu(F.conv3d(x, w, b, (1, 2, 2), (1, 3, 3)))
    _cmp(y, ref)
    g = torch.randn_like(ref)
    for a, e in zip(torch.autograd.grad(y, (w, b), g), torch.autograd.grad(ref, (w, b), g)):
        _cmp(a, e)


@pytest.mark.parametrize("Di,scale", [(5, 4), (7, 2), (1, 3)])
def test_upsample_depth(emu, Di, scale):
    x = _rand(2, 32, Di, 3, 4).requires_grad_()
    y = ops.upsample_depth(x, scale)
    ref = F.interpolate(x, scale_factor=(scale, 1, 1), mode="trilinear", align_corners=True)
    _cmp(y, ref, 1e-6)  # source index is computed in fp32, like ATen does for float tensors
    g = torch.randn_like(ref)
    _cmp(torch.autograd.grad(y, x, g)[0], torch.autograd.grad(ref, x, g)[0], 1e-6)


@pytest.mark.parametrize("Di,scale,K", [(5, 4, (3, 3, 3)), (1, 3, (3, 3, 3)), (6, 2, (5, 3, 3))])
def test_upsample_conv3d_depth(emu, Di, scale, K):
    """interpolate -> conv -> ReLU (ref models/seg_model.py:204-205) as low-resolution (1,kH,kW) conv + tap mixing."""
    x = _rand(2, 32, Di, 5, 6).requires_grad_()
    w = (_rand(16, 32, *K) / (32 * K[0] * K[1] * K[2]) ** 0.5).requires_grad_()
    b = _rand(16).requires_grad_()
    y = ops.upsample_conv3d_depth(x, w, b, scale, act=ops.ACT_RELU)
    up = F.interpolate(x, scale_factor=(scale, 1, 1), mode="trilinear", align_corners=True)
    ref = F.relu(F.conv3d(up, w, b, 1, tuple(k // 2 for k in K)))
    _cmp(y, ref, 1e-5)
    g = torch.randn_like(ref)
    for a, e in zip(torch.autograd.grad(y, [x, w, b], g), torch.autograd.grad(ref, [x, w, b], g)):
        _cmp(a, e, 1e-5)


@pytest.mark.parametrize("Cin,Cout,K,pad", [(32, 2, (1, 1, 1), 0), (16, 2, (5, 5, 5), 2), (32, 16, (3, 3, 3), 1),
                                            (64, 4, (1, 7, 7), (0, 3, 3))])
def test_thin_output_and_half_chunk_convs(emu, Cin, Cout, K, pad):
    x = _rand(2, Cin, 4, 5, 9).requires_grad_()
    w = _rand(Cout, Cin, *K).requires_grad_()
    b = _rand(Cout).requires_grad_()
    y = ops.fused_conv3d(x, w, b, 1, pad, act=ops.ACT_RELU if Cout == 16 else ops.ACT_NONE)
    ref = F.conv3d(x, w, b, 1, pad)
    if Cout == 16:
        ref = torch.relu(ref)
    _cmp(y, ref)
    g = torch.randn_like(ref)
    for a, e in zip(torch.autograd.grad(y, (x, w, b), g), torch.autograd.grad(ref, (x, w, b), g)):
        _cmp(a, e)


def test_choose_tile():
    assert ops.choose_tile((128, 64, 64))[0] * ops.choose_tile((128, 64, 64))[1] * ops.choose_tile((128, 64, 64))[2] == 128
    assert ops.choose_tile((4, 12, 12)) == (0, 0, 0)
    td, th, tw = ops.choose_tile((1, 128, 128))
    assert td == 1 and th * tw == 128


def test_phase_taps_cover_every_tap_once():
    for K, s, p in [(3, 2, 1), (4, 2, 1), (1, 2, 0), (2, 2, 0), (3, 1, 1), (5, 3, 2)]:
        seen = []
        for ph in range(s):
            t = ops.phase_taps(K, s, p, ph)
            if t is None:
                continue
            count, off0, offs, k0, ks = t
            for j in range(count):
                k = k0 + ks * j
                assert 0 <= k < K
                # position s*q+ph reads strided index q+off0+offs*j: check o*s - p + k == s*q+ph
                q = 3
                o = q + off0 + offs * j
                assert o * s - p + k == s * q + ph
                seen.append(k)
        assert sorted(seen) == list(range(K))


def test_split_k_many_depth_taps(emu):
    """feature_fuse-like: one output depth slice, 16 depth taps -> split-K slabs + combine."""
    x = _rand(1, 32, 16, 9, 10).requires_grad_()
    w = _rand(64, 32, 16, 3, 3).requires_grad_()
    b = _rand(64).requires_grad_()
    y = ops.fused_conv3d(x, w, b, 1, (0, 1, 1), act=ops.ACT_LRELU, slope=0.2)
    ref = F.leaky_relu(F.conv3d(x, w, b, 1, (0, 1, 1)), 0.2)
    _cmp(y, ref)
    g = torch.randn_like(ref)
    for a, e in zip(torch.autograd.grad(y, (x, w, b), g), torch.autograd.grad(ref, (x, w, b), g)):
        _cmp(a, e)


def test_split_k_low_resolution_stage(emu):
    """4^3-voxel stage (too small even for the flattened-tile Winograd kernel): the taps are split over (depth, row)
    ranges for forward and input gradient, the combine carries the InstanceNorm statistics."""
    assert ops._tap_split((4, 24, 40), 2, 64, [ops.full_taps(3)] * 3, 128) is None   # Winograd-sized lattice
    halfgrid = ops._tap_split((4, 32, 32), 2, 64, [ops.full_taps(3)] * 3, 128)       # ... that fills 32 of 256 CUs:
    assert halfgrid is not None and [t[0][0] for t in halfgrid] == [1, 1, 1]         # its depth taps as 3 parts
    assert ops._tap_split((4, 12, 12), 32, 512, [ops.full_taps(3)] * 3, 512) is None  # flattened-tile Winograd
    assert ops._tap_split((4, 8, 8), 2, 64, [ops.full_taps(3)] * 3, 128) is not None  # too few tiles for it
    parts = ops._tap_split((4, 4, 4), 2, 64, [ops.full_taps(3)] * 3, 128)
    assert parts is not None and len(parts) == 6
    taps_seen = sorted((t[0][3] + i * t[0][4], t[1][3] + j * t[1][4]) for t in parts
                       for i in range(t[0][0]) for j in range(t[1][0]))
    assert taps_seen == sorted((i, j) for i in range(3) for j in range(3))   # every (kd, kh) exactly once
    x = _rand(2, 128, 4, 4, 4).requires_grad_()
    w = _rand(64, 128, 3, 3, 3).requires_grad_()
    b, ga, be = _rand(64), _rand(64).requires_grad_(), _rand(64).requires_grad_()
    y = ops.fused_conv3d(x, w, b, 1, 1, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01)
    ref = F.leaky_relu(F.instance_norm(F.conv3d(x, w, b, 1, 1), weight=ga, bias=be), 0.01)
    _cmp(y, ref)
    g = torch.randn_like(ref)
    for a, e in zip(torch.autograd.grad(y, (x, w, ga, be), g), torch.autograd.grad(ref, (x, w, ga, be), g)):
        _cmp(a, e)


@pytest.mark.parametrize("K,D,hw", [(16, 4, (5, 6)), (4, 3, (3, 3)), (32, 2, (4, 5))])
def test_uasr_mix(emu, K, D, hw):
    """The fused UASR head (contract of rehr_uasr_mix_*) against the candidate loop of the reference
    (FLAVR_arch.py:203-246, oracle.flavr_oracle.uasr_head), values and all four gradients."""
    N = 2
    om = _rand(N, D * 2 * K, 1, *hw).requires_grad_()
    ue = (2 * _rand(N, D * K, 1, *hw)).requires_grad_()
    wu, bu = _rand(1, K, 1, 1, 1).requires_grad_(), _rand(1).requires_grad_()
    assert ops.uasr_mix_supported(om, ue, D)
    out, unc = ops.uasr_mix(om, ue, wu, bu, D)
    ro, ru = uasr_head(om[:, :, 0], ue[:, :, 0], wu, bu, D)
    _cmp(out, ro, 1e-6)
    _cmp(unc, ru, 1e-6)
    g0, g1 = torch.randn_like(ro), torch.randn_like(ru)
    got = torch.autograd.grad([out, unc], [om, ue, wu, bu], [g0, g1])
    ref = torch.autograd.grad([ro, ru], [om, ue, wu, bu], [g0, g1])
    for a, e in zip(got, ref):
        _cmp(a, e, 1e-6)
    assert not ops.uasr_mix_supported(om, ue[:, :-1], D)
