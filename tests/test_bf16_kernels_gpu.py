"""Mixed-precision kernels (BASELINE.json configs[4]: bf16 operands, fp32 accumulate) against torch.nn.functional
in fp64 evaluated on the SAME bf16-rounded inputs and weights.  Tolerance: what is left is the fp32 accumulation
order and the final rounding of the output to bf16 (2^-9 relative per element) -> 1e-2 of the tensor's max for
bf16 outputs, 1e-4 for fp32 outputs / statistics / weight gradients."""
import pytest
import torch
import torch.nn.functional as F

from rehrseg_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def act(shape, gen, dtype=BF):
    return torch.randn(shape, generator=gen).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last_3d)


def relmax(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-30))


CASES = [  # (N, Cin, Cout, D, H, W, kernel, stride, pad)
    (2, 64, 64, 6, 20, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # K step 64, 64-wide N tile
    (1, 128, 128, 4, 16, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1)),     # 128x128 tile
    (2, 32, 32, 5, 18, 22, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # K step 32, 32-wide N tile
    (1, 32, 64, 8, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1)),       # strided stage entry
    (1, 64, 128, 6, 16, 16, (3, 3, 3), (1, 2, 2), (1, 1, 1)),
    (1, 320, 320, 4, 6, 6, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # 5 chunks of 64
    (1, 48, 96, 4, 10, 12, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # partly empty chunk, padded N
    (2, 64, 64, 4, 12, 12, (1, 1, 1), (1, 1, 1), (0, 0, 0)),       # pointwise
]


@pytest.mark.parametrize("case", CASES)
def test_conv3d_forward_and_input_gradient_bf16(case):
    N, Cin, Cout, D, H, W, K, s, p = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = act((N, Cin, D, H, W), g)
    w = (torch.randn((Cout, Cin) + K, generator=g) / (Cin * K[0] * K[1] * K[2]) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    cfg = ops.ConvCfg(s, p)
    y, stats = ops.conv_forward(x, None, w, b, cfg, ops.ACT_LRELU, 0.1, 2)
    assert y.dtype == BF and y.shape[1] == Cout
    wq = w.to(BF).double()
    ref = F.leaky_relu(F.conv3d(x.double(), wq, b.double(), s, p), 0.1)
    assert relmax(y, ref) < 1e-2
    # statistics come from the fp32 accumulators (before the bf16 rounding of y)
    assert relmax(stats[..., 0], ref.sum((2, 3, 4))) < 2e-3 and relmax(stats[..., 1], (ref ** 2).sum((2, 3, 4))) < 1e-3
    dz = act(tuple(ref.shape), g)
    dx, _ = ops.conv_dgrad(dz, w, (D, H, W), Cin, 0, cfg)
    assert dx.dtype == BF
    xr = x.double().requires_grad_()
    F.conv3d(xr, wq, None, s, p).backward(dz.double())
    assert relmax(dx, xr.grad) < 1e-2


def test_virtual_concat_and_fp32_output_bf16():
    g = torch.Generator().manual_seed(7)
    x1, x2 = act((1, 64, 4, 12, 12), g), act((1, 32, 4, 12, 12), g)
    w = (torch.randn(64, 96, 3, 3, 3, generator=g) / (96 * 27) ** 0.5).to(DEV)
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1))
    y, _ = ops.conv_forward(x1, x2, w, None, cfg, ops.ACT_NONE, 0.0, 0)
    ref = F.conv3d(torch.cat([x1, x2], 1).double(), w.to(BF).double(), None, 1, 1)
    assert relmax(y, ref) < 1e-2
    dz = act(tuple(ref.shape), g)
    d1, d2 = ops.conv_dgrad(dz, w, (4, 12, 12), 64, 32, cfg)
    xr = torch.cat([x1, x2], 1).double().requires_grad_()
    F.conv3d(xr, w.to(BF).double(), None, 1, 1).backward(dz.double())
    assert relmax(d1, xr.grad[:, :64]) < 1e-2 and relmax(d2, xr.grad[:, 64:]) < 1e-2


@pytest.mark.parametrize("K,s,p", [((2, 2, 2), (2, 2, 2), (0, 0, 0)), ((1, 2, 2), (1, 2, 2), (0, 0, 0)),
                                   ((3, 4, 4), (1, 2, 2), (1, 1, 1))])
def test_conv_transpose3d_bf16(K, s, p):
    g = torch.Generator().manual_seed(11 + K[1])
    x = act((1, 128, 4, 8, 8), g)
    w = (torch.randn((128, 64) + K, generator=g) / (128 * K[0]) ** 0.5).to(DEV)
    b = torch.randn(64, generator=g).to(DEV)
    cfg = ops.ConvCfg(s, p, transposed=True)
    y, _ = ops.conv_forward(x, None, w, b, cfg, ops.ACT_NONE, 0.0, 0)
    ref = F.conv_transpose3d(x.double(), w.to(BF).double(), b.double(), s, p)
    assert tuple(y.shape) == tuple(ref.shape) and relmax(y, ref) < 1e-2
    dz = act(tuple(ref.shape), g)
    dx, _ = ops.conv_dgrad(dz, w, tuple(x.shape[2:]), 128, 0, cfg)
    xr = x.double().requires_grad_()
    F.conv_transpose3d(xr, w.to(BF).double(), None, s, p).backward(dz.double())
    assert relmax(dx, xr.grad) < 1e-2


WG_CASES = [  # (N, Cin, Cout, D, H, W, kernel, stride, pad, transposed)
    (2, 64, 64, 6, 20, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1), False),     # 64-tile, waves split K
    (1, 128, 128, 4, 16, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), False),   # 128-tile, 2x2 waves
    (2, 32, 32, 5, 18, 22, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),     # 32-tile
    (1, 32, 64, 8, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1), False),     # strided, mixed channel counts
    (1, 48, 96, 4, 10, 12, (3, 3, 3), (1, 1, 1), (1, 1, 1), False),     # masked tiles
    (1, 128, 64, 4, 8, 8, (2, 2, 2), (2, 2, 2), (0, 0, 0), True),       # ConvTranspose3d, kernel = stride
    (1, 128, 64, 4, 8, 8, (3, 4, 4), (1, 2, 2), (1, 1, 1), True),
]


@pytest.mark.parametrize("case", WG_CASES)
def test_weight_gradient_bf16(case):
    N, Cin, Cout, D, H, W, K, s, p, tr = case
    g = torch.Generator().manual_seed(sum(case[:6]) + 1)
    x = act((N, Cin, D, H, W), g)
    wshape = ((Cin, Cout) if tr else (Cout, Cin)) + K
    w = (torch.randn(wshape, generator=g) * 0.05).to(DEV)
    cfg = ops.ConvCfg(s, p, transposed=tr)
    conv = (lambda a, ww: F.conv_transpose3d(a, ww, None, s, p)) if tr else (lambda a, ww: F.conv3d(a, ww, None, s, p))
    wr = w.double().requires_grad_()
    ref_y = conv(x.double(), wr)
    dz = act(tuple(ref_y.shape), g)
    ref_y.backward(dz.double())
    dw, db = ops.conv_wgrad(dz, x, None, w, cfg, True)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == wshape
    assert relmax(dw, wr.grad) < 1e-4      # exact bf16 products, fp32 accumulation: only the summation order differs
    assert relmax(db, dz.double().sum((0, 2, 3, 4))) < 1e-4


HALO_CASES = [  # shapes the LDS halo-brick kernel takes: (N, Cin, Cout, D, H, W, kernel, pad)
    (1, 32, 32, 8, 16, 32, (3, 3, 3), (1, 1, 1)),      # BN 32: brick 4x8x16, one block per CU
    (2, 32, 32, 5, 32, 32, (1, 3, 3), (0, 1, 1)),      # BN 32: brick 2x16x16? (Ld 5 -> 4x8x16 pads 1.6x) / 1x3x3 taps
    (1, 64, 64, 4, 16, 32, (3, 3, 3), (1, 1, 1)),      # BN 64: brick 2x8x16
    (1, 64, 64, 8, 8, 8, (3, 3, 3), (1, 1, 1)),        # BN 64: brick 4x8x8
    (1, 128, 128, 4, 16, 16, (3, 3, 3), (1, 1, 1)),    # BN 128: brick 2x8x8, 4 chunks
    (1, 48, 96, 6, 16, 32, (3, 3, 3), (1, 1, 1)),      # half-filled last chunk, padded N (96 -> 3 x 32)
    (2, 320, 320, 4, 8, 8, (3, 3, 3), (1, 1, 1)),      # 10 chunks, 5 channel tiles of 64
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_halo_brick_kernel_matches_gather_kernel_and_torch(case):
    from rehrseg_amd import hip_backend
    N, Cin, Cout, D, H, W, K, p = case
    g = torch.Generator().manual_seed(sum(case[:6]) + 3)
    x = act((N, Cin, D, H, W), g)
    w = (torch.randn((Cout, Cin) + K, generator=g) / (Cin * K[0] * 9) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    cfg = ops.ConvCfg((1, 1, 1), p)
    res = {}
    for halo in (True, False):
        hip_backend.USE_HALO_BF16 = halo
        try:
            y, stats = ops.conv_forward(x, None, w, b, cfg, ops.ACT_LRELU, 0.1, 2)
            dz = act(tuple(y.shape), torch.Generator().manual_seed(1))
            dx, _ = ops.conv_dgrad(dz, w, (D, H, W), Cin, 0, cfg)
        finally:
            hip_backend.USE_HALO_BF16 = True
        res[halo] = (y.float(), stats.clone(), dx.float())
    wq = w.to(BF).double()
    ref = F.leaky_relu(F.conv3d(x.double(), wq, b.double(), 1, p), 0.1)
    xr = x.double().requires_grad_()
    F.conv3d(xr, wq, None, 1, p).backward(dz.double())
    for halo in (True, False):
        y, stats, dx = res[halo]
        assert relmax(y, ref) < 1e-2 and relmax(dx, xr.grad) < 1e-2, halo
        assert relmax(stats[..., 0], ref.sum((2, 3, 4))) < 2e-3 and relmax(stats[..., 1], (ref ** 2).sum((2, 3, 4))) < 1e-3
    # the two kernels form the same fp32 sums in a different order: identical after rounding to bf16 almost everywhere
    assert relmax(res[True][0], res[False][0]) < 1e-2 and relmax(res[True][1], res[False][1]) < 1e-5


def test_halo_brick_kernel_virtual_concat_and_transposed_phase_taps():
    g = torch.Generator().manual_seed(21)
    x1, x2 = act((1, 32, 4, 16, 32), g), act((1, 32, 4, 16, 32), g)
    w = (torch.randn(32, 64, 3, 3, 3, generator=g) / (64 * 27) ** 0.5).to(DEV)
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1))
    y, _ = ops.conv_forward(x1, x2, w, None, cfg, ops.ACT_NONE, 0.0, 0)
    ref = F.conv3d(torch.cat([x1, x2], 1).double(), w.to(BF).double(), None, 1, 1)
    assert relmax(y, ref) < 1e-2
    # ConvTranspose3d (3,4,4)/(1,2,2): every output phase is a unit-stride 3x2x2-tap gather written at stride 2
    xt = act((1, 64, 4, 16, 16), g)
    wt = (torch.randn(64, 64, 3, 4, 4, generator=g) / (64 * 12) ** 0.5).to(DEV)
    ct = ops.ConvCfg((1, 2, 2), (1, 1, 1), transposed=True)
    yt, _ = ops.conv_forward(xt, None, wt, None, ct, ops.ACT_NONE, 0.0, 0)
    rt = F.conv_transpose3d(xt.double(), wt.to(BF).double(), None, (1, 2, 2), (1, 1, 1))
    assert relmax(yt, rt) < 1e-2


BRICK_WG_CASES = [  # shapes the LDS-brick weight-gradient kernel takes: (N, Cin, Cout, D, H, W, kernel, pad)
    (1, 32, 32, 8, 16, 32, (3, 3, 3), (1, 1, 1)),
    (2, 64, 32, 4, 16, 32, (3, 3, 3), (1, 1, 1)),      # two channel-tile pairs
    (1, 32, 32, 8, 16, 32, (1, 3, 3), (0, 1, 1)),      # 9 taps
    (1, 48, 96, 6, 16, 16, (3, 3, 3), (1, 1, 1)),      # masked channel tiles, padded bricks along depth
    (2, 128, 128, 4, 8, 16, (3, 3, 3), (1, 1, 1)),     # 16 tile pairs
]


@pytest.mark.parametrize("case", BRICK_WG_CASES)
def test_brick_weight_gradient_bf16_matches_slab_kernel_and_torch(case):
    from rehrseg_amd import hip_backend
    N, Cin, Cout, D, H, W, K, p = case
    g = torch.Generator().manual_seed(sum(case[:6]) + 5)
    x = act((N, Cin, D, H, W), g)
    dz = act((N, Cout, D, H, W), g)
    w = (torch.randn((Cout, Cin) + K, generator=g) * 0.05).to(DEV)
    cfg = ops.ConvCfg((1, 1, 1), p)
    wr = w.double().requires_grad_()
    F.conv3d(x.double(), wr, None, 1, p).backward(dz.double())
    res = {}
    for brick in (True, False):
        hip_backend.USE_WINOGRAD_WGRAD = brick      # False -> REHR_WGRAD_DIRECT: the per-tap slab kernel
        try:
            dw, _ = ops.conv_wgrad(dz, x, None, w, cfg, False)
        finally:
            hip_backend.USE_WINOGRAD_WGRAD = True
        res[brick] = dw
        assert relmax(dw, wr.grad) < 1e-4, brick
    assert relmax(res[True], res[False]) < 1e-5
    # virtual concat: two source tensors, one destination
    if Cin % 64 == 0:
        x1, x2 = x[:, :Cin // 2].contiguous(memory_format=torch.channels_last_3d), x[:, Cin // 2:].contiguous(memory_format=torch.channels_last_3d)
        dw2, _ = ops.conv_wgrad(dz, x1, x2, w, cfg, False)
        assert relmax(dw2, wr.grad) < 1e-4


# ----------------------------------------------------------------------------- sr_head.2 on the matrix cores
THIN5_CASES = [(1, 3, 5, 32), (2, 7, 6, 64), (1, 12, 9, 96), (1, 33, 13, 128), (1, 9, 4, 160)]


@pytest.mark.parametrize("shape", THIN5_CASES)
def test_thin5_conv_vs_torch(shape):
    """Conv3d(16 -> 2, 5x5x5, pad 2) forward / input gradient / weight + bias gradient of the bf16 MFMA kernels
    (models/seg_model.py:199) against torch on the same bf16-rounded operands (fp32 arithmetic).  Products of bf16
    values are exact in fp32, so only the summation order differs: 1e-5 of the largest value; the input gradient is
    stored bf16 (2^-9 relative)."""
    import torch.nn.functional as F
    from rehrseg_amd import hip_backend as hb
    N, D, H, W = shape
    assert hb.thin5_supported((N, 16, D, H, W), (2, 16, 5, 5, 5), (2, 2, 2))
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn(N, 16, D, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(2, 16, 5, 5, 5, generator=g) * 0.05).to(DEV)
    b = torch.randn(2, generator=g).to(DEV)
    dy = torch.randn(N, 2, D, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    xb = x.to(torch.bfloat16)
    xr, wr, dyr = xb.float(), w.to(torch.bfloat16).float(), dy.to(torch.bfloat16).float()

    y = hb.thin5_fwd(xb, w, b)
    ref = F.conv3d(xr, wr, b, padding=2)
    assert y.dtype == torch.float32 and y.shape == ref.shape
    assert (y - ref).abs().max() <= 1e-5 * ref.abs().max()

    dx = hb.thin5_dgrad(dy, w)
    ref = torch.nn.grad.conv3d_input(xr.shape, wr, dyr, padding=2)
    assert dx.dtype == torch.bfloat16
    assert (dx.float() - ref).abs().max() <= 2.0 ** -8 * ref.abs().max()

    dw, db = hb.thin5_wgrad(xb, w, dy, True)
    ref = torch.nn.grad.conv3d_weight(xr, w.shape, dyr, padding=2)
    assert (dw - ref).abs().max() <= 2e-5 * ref.abs().max()
    torch.testing.assert_close(db, dy.sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-3)
    dw2, _ = hb.thin5_wgrad(xb, w, dy, False)
    assert torch.equal(dw, dw2)   # fixed summation order


def test_thin5_rejects_other_shapes():
    from rehrseg_amd import hip_backend as hb, lib
    assert not hb.thin5_supported((1, 16, 8, 8, 48), (2, 16, 5, 5, 5), (2, 2, 2))
    assert not hb.thin5_supported((1, 16, 8, 8, 64), (2, 16, 3, 3, 3), (1, 1, 1))
    x = torch.zeros(1, 16, 8, 8, 48, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    with pytest.raises(lib.RehrsegHipError):
        hb.thin5_fwd(x, torch.zeros(2, 16, 5, 5, 5, device=DEV), None)


def test_upmix_bf16_matches_fp32():
    """The depth-upsample + tap-sum pass with bf16 activations both ways against the fp32 kernel on the same rounded
    inputs (arithmetic is fp32 in both; only the stores round)."""
    from rehrseg_amd import hip_backend as hb
    g_ = torch.Generator(device="cpu").manual_seed(3)
    N, Cc, KD, Di, Do, H, W = 2, 16, 3, 5, 20, 6, 8
    g = torch.randn(N, KD * Cc, Di, H, W, generator=g_).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    bias = torch.randn(Cc, generator=g_).to(DEV)
    gb = g.to(torch.bfloat16)
    y32 = hb.upmix_depth_fwd(gb.float(), bias, Do, Cc, KD, 1, 1, 0.0)
    y16 = hb.upmix_depth_fwd(gb, bias, Do, Cc, KD, 1, 1, 0.0)
    assert y16.dtype == torch.bfloat16
    torch.testing.assert_close(y16.float(), y32, rtol=2.0 ** -8, atol=1e-6)
    dy = torch.randn(N, Cc, Do, H, W, generator=g_).to(DEV).contiguous(memory_format=torch.channels_last_3d).to(torch.bfloat16)
    dg32 = hb.upmix_depth_bwd(dy.float(), y16.float(), Di, KD, 1, 1, 0.0)
    dg16 = hb.upmix_depth_bwd(dy, y16, Di, KD, 1, 1, 0.0)
    assert dg16.dtype == torch.bfloat16
    torch.testing.assert_close(dg16.float(), dg32, rtol=2.0 ** -8, atol=1e-5)
    db32 = hb.channel_sum_actgrad(dy.float(), y16.float(), 1, 0.0)
    db16 = hb.channel_sum_actgrad(dy, y16, 1, 0.0)
    torch.testing.assert_close(db16, db32, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("shape", [(1, 3, 5, 32), (2, 7, 6, 64), (1, 12, 9, 96), (1, 33, 13, 128)])
def test_thin5_conv_fp32_vs_fp64(shape):
    """The same layer on the fp32 matrix cores (thin_conv_f32.hip; the fp32 path's sr_head.2) against torch in float64
    on the host: forward, input gradient, weight + bias gradient within 1e-5 of the largest value."""
    import torch.nn.functional as F
    from rehrseg_amd import hip_backend as hb
    N, D, H, W = shape
    assert hb.thin5_supported((N, 16, D, H, W), (2, 16, 5, 5, 5), (2, 2, 2), torch.float32)
    assert not hb.thin5_supported((N, 16, D, H, 160), (2, 16, 5, 5, 5), (2, 2, 2), torch.float32)
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + 1)
    x = torch.randn(N, 16, D, H, W, generator=g).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(2, 16, 5, 5, 5, generator=g) * 0.05
    b = torch.randn(2, generator=g)
    dy = torch.randn(N, 2, D, H, W, generator=g).contiguous(memory_format=torch.channels_last_3d)
    xd, wd, dyd = x.double(), w.double(), dy.double()

    def close(a, ref, tol):
        assert a.dtype == torch.float32
        assert (a.double().cpu() - ref).abs().max() <= tol * ref.abs().max()

    close(hb.thin5_fwd(x.to(DEV), w.to(DEV), b.to(DEV)), F.conv3d(xd, wd, b.double(), padding=2), 1e-5)
    close(hb.thin5_dgrad(dy.to(DEV), w.to(DEV), torch.float32), torch.nn.grad.conv3d_input(xd.shape, wd, dyd, padding=2), 1e-5)
    dw, db = hb.thin5_wgrad(x.to(DEV), w.to(DEV), dy.to(DEV), True)
    close(dw, torch.nn.grad.conv3d_weight(xd, w.shape, dyd, padding=2), 1e-5)
    close(db, dyd.sum((0, 2, 3, 4)), 1e-5)


def test_sr_head_fp32_routes_to_matrix_cores():
    """fused_conv3d on the sr_head.2 shape in fp32: forward and both gradients agree with the VALU kernels they replace."""
    from rehrseg_amd import hip_backend as hb
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(1, 16, 9, 6, 64, generator=g).to(DEV).requires_grad_()
    w = (torch.randn(2, 16, 5, 5, 5, generator=g) * 0.05).to(DEV).requires_grad_()
    b = torch.randn(2, generator=g).to(DEV).requires_grad_()
    gy = torch.randn(1, 2, 9, 6, 64, generator=g).to(DEV)
    y = ops.fused_conv3d(x, w, b, 1, 2)
    got = torch.autograd.grad(y, (x, w, b), gy)
    saved = hb.thin5_supported
    hb.thin5_supported = lambda *a, **k: False          # the VALU kernels of direct_conv.hip
    try:
        y2 = ops.fused_conv3d(x, w, b, 1, 2)
        want = torch.autograd.grad(y2, (x, w, b), gy)
    finally:
        hb.thin5_supported = saved
    torch.testing.assert_close(y, y2, rtol=1e-5, atol=1e-5)
    for a, e in zip(got, want):
        torch.testing.assert_close(a, e, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("Cin,Cout,K,stride,pad", [(1, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)), (1, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                                                    (2, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3))])
def test_thin_input_layers_store_bf16_directly(Cin, Cout, K, stride, pad):
    """Mixed precision: the thin-input layers compute in fp32 on the fp32 image and store bf16 (no cast pass).  The
    arithmetic is the fp32 kernel's, so the stored values are exactly its results rounded to bf16, the statistics are
    identical, and the weight gradient from a bf16 dY equals the fp32 kernel's on the same values."""
    from rehrseg_amd import hip_backend as hb
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, Cin, 5, 40, 70, generator=g).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(Cout, Cin, *K, generator=g) / (Cin * K[0] * K[1] * K[2]) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    cfg = ops.ConvCfg(stride, pad, False)
    y32, st32 = ops.conv_forward(x, None, w, b, cfg, ops.ACT_RELU, 0.0, 2)
    with ops.mixed_precision():
        y16, st16 = ops.conv_forward(x, None, w, b, cfg, ops.ACT_RELU, 0.0, 2)
    assert y16.dtype == torch.bfloat16 and y32.dtype == torch.float32
    assert torch.equal(y16, y32.to(torch.bfloat16))
    assert torch.equal(st16, st32)
    dy = torch.randn(y32.shape, generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    assert hb.small_cin_wgrad_on_mfma(x, w, dy, stride, pad)
    dw16, db16 = ops.conv_wgrad(dy, x, None, w, cfg, True)
    dw32, db32 = ops.conv_wgrad(dy.float(), x, None, w, cfg, True)
    assert torch.equal(dw16, dw32) and torch.equal(db16, db32)


def test_teacher_window_stem_in_mixed_precision_stores_bf16():
    """encoder_on_windows under mixed_precision(): the per-slice stem responses stay fp32 (they are combined linearly),
    the assembled windows are stored as bf16 -- exactly the fp32 assembly rounded."""
    import torch.nn.functional as Fn
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    from rehrseg_amd.models.FLAVR.resnet_3D import encoder_on_windows
    torch.manual_seed(5)
    enc = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True).to(DEV).eval().encoder
    B, D, H, W = 1, 7, 32, 48
    xp = Fn.pad(torch.randn(B, 2, D, H, W, device=DEV), (0, 0, 0, 0, 1, 2))
    with torch.no_grad():
        f32 = encoder_on_windows(enc, xp[:, :, :D + 2], D - 1, 0)[0]
        with ops.mixed_precision():
            f16 = encoder_on_windows(enc, xp[:, :, :D + 2], D - 1, 0)[0]
    assert f16.dtype == torch.bfloat16 and torch.equal(f16, f32.to(torch.bfloat16))


@pytest.mark.parametrize("stats", [False, True])
def test_split_k_in_mixed_precision(stats):
    """Low-resolution stage of the nnU-Net plans in bf16 (320 -> 320 at 8 x 5 x 5: a handful of lattice tiles, 8640
    products per output): the taps run as parts of one gather grid into fp32 slabs, the combine stores bf16 and forms
    the statistics from the fp32 sums.  Against fp64 on the same bf16-rounded operands, forward and input gradient."""
    g = torch.Generator().manual_seed(41)
    x = torch.randn(1, 320, 8, 5, 5, generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(320, 320, 3, 3, 3, generator=g) / (320 * 27) ** 0.5).to(DEV)
    b = torch.randn(320, generator=g).to(DEV)
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    assert ops._tap_split((8, 5, 5), 1, 320, [ops.full_taps(3)] * 3, 320, True) is not None
    y, st = ops.conv_forward(x, None, w, b, cfg, ops.ACT_NONE, 0.0, 2 if stats else 0)
    assert y.dtype == torch.bfloat16
    ref = F.conv3d(x.double().cpu(), w.to(torch.bfloat16).double().cpu(), b.double().cpu(), 1, 1)
    assert relmax(y, ref) < 2.0 ** -7
    if stats:
        want = torch.stack([ref.sum((2, 3, 4)), (ref * ref).sum((2, 3, 4))], -1)
        torch.testing.assert_close(st.cpu(), want, rtol=1e-4, atol=1e-3)
    dz = torch.randn(1, 320, 8, 5, 5, generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    dx = ops.conv_dgrad(dz, w, (8, 5, 5), 320, 0, cfg)[0]
    xr = x.double().cpu().requires_grad_()
    F.conv3d(xr, w.to(torch.bfloat16).double().cpu(), None, 1, 1).backward(dz.double().cpu())
    assert dx.dtype == torch.bfloat16 and relmax(dx, xr.grad) < 2.0 ** -7


def test_instnorm_backward_carries_the_conv_bias_gradient():
    """rehr_instnorm_act_bwd_dbias_bf16: the apply pass also returns the column sums of its dx (the gradient of the conv
    bias in front of the normalisation), summed from the fp32 values BEFORE the bf16 store -- like the fp32 path, where
    this gradient is the analytic zero plus fp32 rounding.  Checked against the column sums of the fp32 kernel's dx on the
    same operands; dx / dgamma / dbeta are those of the plain call."""
    from rehrseg_amd import hip_backend as hb
    g = torch.Generator().manual_seed(77)
    N, Cc, D, H, W = 2, 96, 5, 18, 14
    x = act((N, Cc, D, H, W), g)
    dy = act((N, Cc, D, H, W), g)
    gamma = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(Cc, generator=g) * 0.1).to(DEV)
    stats = torch.zeros((N, Cc, 2), dtype=torch.float64, device=DEV)
    xf = x.double()
    stats[..., 0] = xf.sum((2, 3, 4))
    stats[..., 1] = (xf * xf).sum((2, 3, 4))
    _, mr = hb.instnorm_act_fwd(x, stats, gamma, beta, 1e-5, ops.ACT_LRELU, 0.01)
    dx0, dg0, db0 = hb.instnorm_act_bwd(dy, x, mr, gamma, beta, ops.ACT_LRELU, 0.01)
    dx1, dg1, db1, dcb = hb.instnorm_act_bwd(dy, x, mr, gamma, beta, ops.ACT_LRELU, 0.01, want_conv_bias=True)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    dx32, _, _ = hb.instnorm_act_bwd(dy.float(), x.float(), mr, gamma, beta, ops.ACT_LRELU, 0.01)
    assert relmax(dx1, dx32) < 2.0 ** -7                               # the same arithmetic, another store
    want = dx32.double().sum((0, 2, 3, 4))
    scale = float(dx32.double().abs().sum((0, 2, 3, 4)).max())
    assert float((dcb.double() - want).abs().max()) <= 5e-6 * scale
