"""Parity of the HIP kernels (through the C-ABI) against torch.nn.functional on
the CPU in fp64.  Tolerance: 1e-4 of the reference's max magnitude for forward
values and gradients (fp32 MFMA is an exact-fp32 fma chain; only the summation
order differs from ATen/oneDNN)."""
import pytest
import torch
import torch.nn.functional as F

from rehrseg_amd import ops

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _close(got, ref, tol=TOL):
    got = got.detach().double().cpu()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = float(ref.abs().max()) + 1e-30
    err = float((got - ref).abs().max()) / scale
    assert err <= tol, f"rel err {err:.3e} > {tol}"


def _mk(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def _run(fn_gpu, fn_ref, tensors, grad_mask):
    dev = _dev()
    gin = [t.to(dev).requires_grad_(m) for t, m in zip(tensors, grad_mask)]
    rin = [t.double().requires_grad_(m) for t, m in zip(tensors, grad_mask)]
    y = fn_gpu(*gin)
    ref = fn_ref(*rin)
    _close(y, ref)
    g = _mk(*ref.shape, seed=99)
    gg = torch.autograd.grad(y, [t for t, m in zip(gin, grad_mask) if m], g.to(dev))
    rg = torch.autograd.grad(ref, [t for t, m in zip(rin, grad_mask) if m], g.double())
    for a, e in zip(gg, rg):
        _close(a, e)


CONVS = [
    (64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 5, 12, 20)),
    (64, 128, (3, 3, 3), (1, 2, 2), (1, 1, 1), (1, 4, 18, 16)),
    (64, 128, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 3, 16, 14)),
    (128, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 4, 8, 8)),
    (32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 9, 11)),
    (32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 9, 10, 11)),
    (320, 320, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 4, 4, 4)),
    (32, 32, (5, 5, 5), (1, 1, 1), (2, 2, 2), (1, 6, 7, 8)),
]


@pytest.mark.parametrize("Cin,Cout,K,stride,pad,dims", CONVS)
def test_conv3d(Cin, Cout, K, stride, pad, dims):
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=1)
    w = _mk(Cout, Cin, *K, seed=2) / (Cin * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(Cout, seed=3)
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, stride, pad, act=ops.ACT_LRELU, slope=0.2),
         lambda x, w, b: F.leaky_relu(F.conv3d(x, w, b, stride, pad), 0.2), [x, w, b], [True, True, True])


HALO = [  # shapes the halo-tile kernel takes (stride 1, Cout <= 64, brick-friendly extents)
    (64, 64, (3, 3, 3), (1, 1, 1), (1, 4, 16, 24)),
    (32, 32, (3, 3, 3), (1, 1, 1), (2, 2, 8, 16)),
    (96, 64, (3, 3, 3), (1, 1, 1), (1, 6, 16, 16)),
    (16, 32, (3, 3, 3), (1, 1, 1), (1, 4, 8, 8)),
    (32, 16, (3, 3, 3), (1, 1, 1), (1, 4, 16, 8)),
    (32, 32, (1, 3, 3), (0, 1, 1), (2, 2, 16, 16)),
    (64, 64, (3, 3, 3), (1, 1, 1), (1, 5, 17, 23)),   # ragged edges: masked rows
]


@pytest.fixture
def no_winograd(monkeypatch):
    """The Winograd kernels take unit-stride 3x3 taps first; switch them off to reach the
    direct kernels behind them."""
    from rehrseg_amd import hip_backend
    monkeypatch.setattr(hip_backend, "USE_WINOGRAD", False)


WINO = [  # Cin, Cout, K, pad, dims, expected kernel launches through the Winograd path (fwd + dgrad)
    (64, 64, (3, 3, 3), (1, 1, 1), (1, 4, 16, 32)),     # big tile (64 tiles x 64 channels)
    (48, 64, (3, 3, 3), (1, 1, 1), (2, 3, 32, 16)),     # half last chunk
    (64, 128, (3, 3, 3), (1, 1, 1), (1, 3, 30, 31)),    # ragged edges: masked tiles
    (32, 32, (3, 3, 3), (1, 1, 1), (1, 4, 16, 16)),     # 32 channels: small tile kernel
    (64, 96, (3, 3, 3), (1, 1, 1), (1, 2, 8, 16)),      # Npad 96, 8-row lattice: small tile kernel
    (32, 64, (1, 3, 3), (0, 1, 1), (2, 3, 16, 16)),     # a single depth tap
    (128, 64, (3, 3, 3), (1, 1, 1), (1, 1, 16, 16)),    # depth 1: both outer depth taps fall outside
    (32, 32, (3, 3, 3), (1, 1, 1), (1, 3, 32, 16)),     # wide tile (128 tiles x 32 channels), exact fit
    (48, 32, (3, 3, 3), (1, 1, 1), (2, 2, 64, 32)),     # wide tile: 3 half chunks, 2 x 2 regions
    (32, 96, (3, 3, 3), (1, 1, 1), (1, 2, 60, 30)),     # wide tile: 3 channel tiles, ragged edges
    (64, 32, (1, 3, 3), (0, 1, 1), (1, 2, 32, 48)),     # wide tile, one depth tap
    (128, 128, (3, 3, 3), (1, 1, 1), (32, 4, 12, 12)),  # small planes: flattened tiles, 36 per slice, 3 slices / block
    (128, 128, (3, 3, 3), (1, 1, 1), (64, 4, 8, 8)),    # flattened tiles, 16 per slice (4 whole slices per block)
    (128, 128, (3, 3, 3), (1, 1, 1), (42, 3, 14, 10)),  # flattened tiles, 35 per slice, last block partly empty
    (128, 128, (3, 3, 3), (1, 1, 1), (40, 3, 13, 11)),  # flattened tiles, odd extents: half-empty edge tiles
    (48, 128, (1, 3, 3), (0, 1, 1), (120, 2, 12, 12)),  # flattened tiles, one depth tap, half last chunk
]


@pytest.mark.parametrize("Cin,Cout,K,pad,dims", WINO)
def test_winograd_conv(Cin, Cout, K, pad, dims):
    from rehrseg_amd import hip_backend
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=70)
    w = _mk(Cout, Cin, *K, seed=71) / (Cin * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(Cout, seed=72)
    before = hip_backend.wino_launches
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, 1, pad, act=ops.ACT_LRELU, slope=0.1),
         lambda x, w, b: F.leaky_relu(F.conv3d(x, w, b, 1, pad), 0.1), [x, w, b], [True, True, True])
    assert hip_backend.wino_launches - before == 2, "forward and input gradient should both take the Winograd path"


FLAT8 = [  # tiles per block (0 = the library's pick, 1 = 32, 2 = 64), Cin, Cout, K, pad, dims
    (1, 128, 128, (3, 3, 3), (1, 1, 1), (32, 4, 12, 12)),   # the reference's layer3 planes, two 32-tile blocks per CU
    (2, 128, 128, (3, 3, 3), (1, 1, 1), (32, 4, 12, 12)),   # ... one 64-tile block per CU
    (0, 32, 128, (3, 3, 3), (1, 1, 1), (7, 5, 24, 24)),    # layer2 planes (rows of 12 tiles), last block partly empty
    (1, 48, 128, (3, 3, 3), (1, 1, 1), (40, 3, 13, 11)),   # odd extents, half last chunk
    (2, 32, 192, (1, 3, 3), (0, 1, 1), (30, 2, 20, 28)),   # one depth tap, 3 channel tiles, 14-tile rows
    (2, 32, 128, (3, 3, 3), (1, 1, 1), (128, 1, 12, 12)),   # depth 1: the outer depth taps are never reachable
]


@pytest.mark.parametrize("tiles,Cin,Cout,K,pad,dims", FLAT8)
def test_winograd_flat8_conv(tiles, Cin, Cout, K, pad, dims, monkeypatch):
    """wino_flat8_conv_kernel<1|2> (flattened tiles, row-range staging) forward + input gradient against fp64."""
    from rehrseg_amd import hip_backend
    monkeypatch.setattr(hip_backend, "WINO_FLAT8_TILES", tiles)
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=170)
    w = _mk(Cout, Cin, *K, seed=171) / (Cin * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(Cout, seed=172)
    before = hip_backend.wino_launches
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, 1, pad, act=ops.ACT_LRELU, slope=0.1),
         lambda x, w, b: F.leaky_relu(F.conv3d(x, w, b, 1, pad), 0.1), [x, w, b], [True, True, True])
    # (the input gradient of the narrow layers has too few 64-channel units for this kernel and goes elsewhere)
    assert hip_backend.wino_launches - before >= (2 if Cin >= 128 else 1)


@pytest.mark.parametrize("tiles", [1, 2])
def test_winograd_flat8_statistics_and_concat(tiles, monkeypatch):
    """The per-(slice slot, channel) block sums of the flattened-tile kernel (SE mean, InstanceNorm mean / variance:
    a block's tiles belong to up to three samples) and the two-source input of a decoder conv."""
    from rehrseg_amd import hip_backend
    monkeypatch.setattr(hip_backend, "WINO_FLAT8_TILES", tiles)
    x = _mk(32, 64, 4, 12, 12, seed=180)
    x2 = _mk(32, 32, 4, 12, 12, seed=181)
    w = _mk(128, 96, 3, 3, 3, seed=182) / (96 * 27) ** 0.5
    b = _mk(128, seed=183)
    aw, ab = _mk(128, 128, 1, 1, 1, seed=184) / 11.0, _mk(128, seed=185)
    _run(lambda x, x2, w, b, aw, ab: ops.fused_conv3d(x, w, b, 1, 1, x2=x2, se=(aw, ab), act=ops.ACT_LRELU, slope=0.2),
         lambda x, x2, w, b, aw, ab: F.leaky_relu(_se(F.conv3d(torch.cat([x, x2], 1), w, b, 1, 1), aw, ab), 0.2),
         [x, x2, w, b, aw, ab], [True] * 6)
    ga, be = _mk(128, seed=186), _mk(128, seed=187)
    _run(lambda x, x2, w, b, ga, be: ops.fused_conv3d(x, w, b, 1, 1, x2=x2, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01),
         lambda x, x2, w, b, ga, be: F.leaky_relu(F.instance_norm(F.conv3d(torch.cat([x, x2], 1), w, b, 1, 1),
                                                                  weight=ga, bias=be), 0.01),
         [x, x2, w, b, ga, be], [True, True, True, False, True, True])


def test_small_planes_without_the_flattened_tile_kernel(monkeypatch):
    """REHR_DBG_GG_NO_FLAT8: the 12 x 12 planes fall back to the region / direct kernels -- same results."""
    from rehrseg_amd import hip_backend
    monkeypatch.setattr(hip_backend, "USE_WINO_FLAT8", False)
    x = _mk(32, 64, 4, 12, 12, seed=190)
    w = _mk(128, 64, 3, 3, 3, seed=191) / (64 * 27) ** 0.5
    b = _mk(128, seed=192)
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, 1, 1, act=ops.ACT_RELU),
         lambda x, w, b: torch.relu(F.conv3d(x, w, b, 1, 1)), [x, w, b], [True, True, True])


WINO_WGRAD = [  # weight gradients in the transform domain: >= 64 channels both sides, W >= 16
    (64, 64, (3, 3, 3), (1, 1, 1), (1, 4, 16, 32)),
    (128, 64, (3, 3, 3), (1, 1, 1), (2, 3, 8, 16)),
    (64, 192, (3, 3, 3), (1, 1, 1), (1, 2, 14, 31)),     # ragged region edges, 3 output-channel tiles
    (80, 64, (1, 3, 3), (0, 1, 1), (2, 3, 16, 16)),      # channel tail (80 -> 128 padded is refused: direct path)
    (64, 64, (3, 3, 3), (1, 1, 1), (3, 2, 4, 16)),       # one region per slice, three samples
    (32, 32, (3, 3, 3), (1, 1, 1), (2, 3, 16, 16)),      # 32-channel tiles (1 x 1 groups)
    (64, 32, (3, 3, 3), (1, 1, 1), (1, 4, 8, 32)),       # 1 x 2 groups
    (32, 64, (3, 3, 3), (1, 1, 1), (1, 4, 12, 16)),      # 2 x 1 groups
    (32, 16, (3, 3, 3), (1, 1, 1), (2, 4, 16, 32)),      # 16 output channels padded to one 32-wide group (SR head)
    (64, 64, (3, 3, 3), (1, 1, 1), (16, 4, 12, 12)),     # narrow planes: 8 slices side by side in a virtual lattice
    (128, 64, (1, 3, 3), (0, 1, 1), (52, 2, 8, 8)),      # narrow planes, 104 slices = 13 groups, one depth tap
    (64, 128, (3, 3, 3), (1, 1, 1), (63, 1, 12, 10)),    # narrow planes, 63 slices (last group partial), width 10
]


@pytest.mark.parametrize("Cin,Cout,K,pad,dims", WINO_WGRAD)
def test_winograd_wgrad(Cin, Cout, K, pad, dims):
    from rehrseg_amd import hip_backend
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=80)
    w = _mk(Cout, Cin, *K, seed=81) / (Cin * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(Cout, seed=82)
    before = hip_backend.wino_wgrad_launches
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, 1, pad), lambda x, w, b: F.conv3d(x, w, b, 1, pad),
         [x, w, b], [True, True, True])
    expect = 0 if Cin == 80 else 1
    assert hip_backend.wino_wgrad_launches - before == expect


def test_winograd_virtual_concat_instnorm():
    from rehrseg_amd import hip_backend
    x1, x2 = _mk(2, 32, 3, 16, 16, seed=73), _mk(2, 64, 3, 16, 16, seed=74)
    w = _mk(64, 96, 3, 3, 3, seed=75) / 51.0
    b, ga, be = _mk(64, seed=76), _mk(64, seed=77), _mk(64, seed=78)
    before = hip_backend.wino_launches
    _run(lambda a, c, w, b, ga, be: ops.fused_conv3d(a, w, b, 1, 1, x2=c, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01),
         lambda a, c, w, b, ga, be: F.leaky_relu(
             F.instance_norm(F.conv3d(torch.cat([a, c], 1), w, b, 1, 1), weight=ga, bias=be), 0.01),
         [x1, x2, w, b, ga, be], [True, True, True, False, True, True])
    assert hip_backend.wino_launches > before


@pytest.mark.parametrize("Cin,Cout,K,pad,dims", HALO)
def test_halo_tile_conv(Cin, Cout, K, pad, dims, no_winograd):
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=50)
    w = _mk(Cout, Cin, *K, seed=51) / (Cin * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(Cout, seed=52)
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, 1, pad, act=ops.ACT_RELU),
         lambda x, w, b: torch.relu(F.conv3d(x, w, b, 1, pad)), [x, w, b], [True, True, True])


def test_halo_tile_conv_virtual_concat_instnorm(no_winograd):
    x1, x2 = _mk(2, 32, 4, 16, 16, seed=53), _mk(2, 32, 4, 16, 16, seed=54)
    w = _mk(32, 64, 3, 3, 3, seed=55) / 41.0
    b, ga, be = _mk(32, seed=56), _mk(32, seed=57), _mk(32, seed=58)
    _run(lambda a, c, w, b, ga, be: ops.fused_conv3d(a, w, b, 1, 1, x2=c, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01),
         lambda a, c, w, b, ga, be: F.leaky_relu(
             F.instance_norm(F.conv3d(torch.cat([a, c], 1), w, b, 1, 1), weight=ga, bias=be), 0.01),
         [x1, x2, w, b, ga, be], [True, True, True, False, True, True])


def test_split_k_many_depth_taps():
    """feature_fuse-like contraction (one output slice, 32 depth taps): split-K slabs + combine."""
    x = _mk(1, 64, 32, 24, 20, seed=60)
    w = _mk(64, 64, 32, 3, 3, seed=61) / (64 * 32 * 9) ** 0.5
    b = _mk(64, seed=62)
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, 1, (0, 1, 1), act=ops.ACT_LRELU, slope=0.2),
         lambda x, w, b: F.leaky_relu(F.conv3d(x, w, b, 1, (0, 1, 1)), 0.2), [x, w, b], [True, True, True])


def test_split_k_many_depth_taps_winograd():
    """The same contraction on a plane the Winograd kernels accept (32 x 32, 64 channels): the 8 tap-range parts
    (4 depth taps each) run as ONE launch of the big-tile kernel, every part with its own transformed weights and
    slab; the input gradient (one valid depth tap per output slice) and the weight gradient (32 depth taps = 32
    grid.z slices) take the Winograd kernels too."""
    from rehrseg_amd import hip_backend
    x = _mk(1, 64, 32, 32, 32, seed=70)
    w = _mk(64, 64, 32, 3, 3, seed=71) / (64 * 32 * 9) ** 0.5
    b = _mk(64, seed=72)
    before, before_w = hip_backend.wino_launches, hip_backend.wino_wgrad_launches
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, 1, (0, 1, 1), act=ops.ACT_LRELU, slope=0.2),
         lambda x, w, b: F.leaky_relu(F.conv3d(x, w, b, 1, (0, 1, 1)), 0.2), [x, w, b], [True, True, True])
    assert hip_backend.wino_launches - before == 8 + 1      # 8 forward parts + the input gradient
    assert hip_backend.wino_wgrad_launches - before_w == 1


def test_split_k_strided_low_resolution_stage():
    """nnU-Net's last down-sampling convs (stride 2 into 8^3 / 4^3 voxels, 256-320 channels): tap ranges in one grid."""
    for Cin, Cout, D in ((256, 320, 16), (320, 320, 8)):
        x = _mk(2, Cin, D, D, D, seed=271)
        w = _mk(Cout, Cin, 3, 3, 3, seed=272) / (Cin * 27) ** 0.5
        b, ga, be = _mk(Cout, seed=273), _mk(Cout, seed=274), _mk(Cout, seed=275)
        _run(lambda x, w, b, ga, be: ops.fused_conv3d(x, w, b, 2, 1, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01),
             lambda x, w, b, ga, be: F.leaky_relu(F.instance_norm(F.conv3d(x, w, b, 2, 1), weight=ga, bias=be), 0.01),
             [x, w, b, ga, be], [True, True, False, True, True])


def test_winograd_depth_tap_split_half_filled_grid():
    """nnU-Net's 16^3 stage (2 x 256 x 16^3: 128 big-tile Winograd blocks on 256 CUs): the three depth taps as three
    parts of one Winograd grid + the combine (bias, InstanceNorm statistics), forward and input gradient."""
    from rehrseg_amd import hip_backend
    x = _mk(2, 128, 16, 16, 16, seed=263)
    w = _mk(128, 128, 3, 3, 3, seed=264) / (128 * 27) ** 0.5
    b, ga, be = _mk(128, seed=265), _mk(128, seed=266), _mk(128, seed=267)
    assert ops._tap_split((16, 16, 16), 2, 128, [ops.full_taps(3)] * 3, 128) is not None
    before = hip_backend.wino_launches
    _run(lambda x, w, b, ga, be: ops.fused_conv3d(x, w, b, 1, 1, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01),
         lambda x, w, b, ga, be: F.leaky_relu(F.instance_norm(F.conv3d(x, w, b, 1, 1), weight=ga, bias=be), 0.01),
         [x, w, b, ga, be], [True, True, False, True, True])
    assert hip_backend.wino_launches - before == 6   # 3 forward parts + 3 input-gradient parts


def test_split_k_low_resolution_stage_with_instnorm():
    """nnU-Net bottom stage (4^3 voxels, hundreds of channels): 6 tap ranges in one grid, the combine carries
    bias + InstanceNorm statistics; the input gradient takes the same route.  (8^3 stages go to the
    flattened-tile Winograd kernel.)"""
    x = _mk(2, 128, 4, 4, 4, seed=63)
    w = _mk(192, 128, 3, 3, 3, seed=64) / (128 * 27) ** 0.5
    b, ga, be = _mk(192, seed=65), _mk(192, seed=66), _mk(192, seed=67)
    _run(lambda x, w, b, ga, be: ops.fused_conv3d(x, w, b, 1, 1, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01),
         lambda x, w, b, ga, be: F.leaky_relu(F.instance_norm(F.conv3d(x, w, b, 1, 1), weight=ga, bias=be), 0.01),
         [x, w, b, ga, be], [True, True, False, True, True])


TCONVS = [
    (128, 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), (1, 4, 9, 10)),
    (64, 32, (2, 2, 2), (2, 2, 2), (0, 0, 0), (2, 3, 5, 6)),
    (320, 320, (1, 2, 2), (1, 2, 2), (0, 0, 0), (1, 4, 4, 4)),
    (64, 32, (3, 4, 4), (1, 2, 2), (1, 1, 1), (1, 4, 16, 16)),   # stride phases through the halo-tile kernel
    (128, 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), (2, 2, 8, 24)),
]


@pytest.mark.parametrize("Cin,Cout,K,stride,pad,dims", TCONVS)
def test_conv_transpose3d(Cin, Cout, K, stride, pad, dims):
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=4)
    w = _mk(Cin, Cout, *K, seed=5) / (Cin * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(Cout, seed=6)
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, stride, pad, transposed=True),
         lambda x, w, b: F.conv_transpose3d(x, w, b, stride, pad), [x, w, b], [True, True, True])


WINO22 = [  # transposed convs whose 2x2-tap stride phases / stride-2 input gradient take the F(2x2,2x2) kernel
    (64, 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), (1, 3, 16, 16)),
    (128, 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), (2, 2, 30, 32)),   # ragged region edges
    (96, 128, (1, 4, 4), (1, 2, 2), (0, 1, 1), (1, 2, 16, 32)),   # one depth tap, 2 channel tiles
]


WINO22_FLAT = [  # small planes: the flattened-tile form of the F(2x2,2x2) kernel (forward phases + stride-2 input gradient)
    (128, 128, (3, 4, 4), (1, 2, 2), (1, 1, 1), (32, 8, 12, 12)),  # the reference's 12 x 12 planes: 36 tiles per slice
    (64, 128, (3, 4, 4), (1, 2, 2), (1, 1, 1), (12, 6, 24, 24)),   # 24 x 24 planes, rows of 12 tiles
    (64, 128, (1, 4, 4), (1, 2, 2), (0, 1, 1), (56, 4, 13, 11)),   # odd extents, one depth tap, two channel tiles
    (96, 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), (32, 4, 12, 12)),    # one channel tile: only the 4-phase grid fills the chip
]


@pytest.mark.parametrize("Cin,Cout,K,stride,pad,dims", WINO22_FLAT)
def test_winograd22_flat_conv_transpose(Cin, Cout, K, stride, pad, dims):
    """wino22_flat_conv_kernel behind ConvTranspose3d (4 output phases, SE statistics over blocks that span several
    samples) and behind its input gradient (stride-2, 4-tap gather: four source parities), against fp64."""
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=290)
    w = _mk(Cin, Cout, *K, seed=291) / (Cin * K[0] * 4) ** 0.5
    b = _mk(Cout, seed=292)
    aw, ab = _mk(Cout, Cout, 1, 1, 1, seed=293) / Cout ** 0.5, _mk(Cout, seed=294)
    # (no activation here: with 10^7 outputs a few always sit within rounding of LeakyReLU's kink, and one flipped branch
    # is a 2 % error of the input gradient's max norm in fp32 AND in the direct kernels -- not what this test is about)
    _run(lambda x, w, b, aw, ab: ops.fused_conv3d(x, w, b, stride, pad, transposed=True, se=(aw, ab)),
         lambda x, w, b, aw, ab: _se(F.conv_transpose3d(x, w, b, stride, pad), aw, ab),
         [x, w, b, aw, ab], [True] * 5)
    if Cin == 96:   # the decoder's two-source form (virtual concat), phases in one grid
        xa_, xb_ = x[:, :64].contiguous(), x[:, 64:].contiguous()
        _run(lambda xa_, xb_, w, b: ops.fused_conv3d(xa_, w, b, stride, pad, x2=xb_, transposed=True),
             lambda xa_, xb_, w, b: F.conv_transpose3d(torch.cat([xa_, xb_], 1), w, b, stride, pad),
             [xa_, xb_, w, b], [True] * 4)


@pytest.mark.parametrize("Cin,Cout,K,stride,pad,dims", WINO22)
def test_winograd22_conv_transpose(Cin, Cout, K, stride, pad, dims):
    from rehrseg_amd import hip_backend
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=90)
    w = _mk(Cin, Cout, *K, seed=91) / (Cin * K[0] * 4) ** 0.5
    b = _mk(Cout, seed=92)
    before, before_w = hip_backend.wino_launches, hip_backend.wino_wgrad_launches
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, stride, pad, transposed=True, act=ops.ACT_LRELU, slope=0.2),
         lambda x, w, b: F.leaky_relu(F.conv_transpose3d(x, w, b, stride, pad), 0.2), [x, w, b], [True, True, True])
    assert hip_backend.wino_wgrad_launches - before_w == 1, "the weight gradient takes the F(2x2,2x2) kernel too"
    # 4 forward phases + the stride-2 input gradient (whose 96 output channels pad to 96, not a 64-multiple)
    assert hip_backend.wino_launches - before == (4 if Cin == 96 else 5)


def test_winograd22_virtual_concat_se():
    """FLAVR upConv3D on a skip concat with the SE statistics epilogue (decoder.{1,2,4})."""
    x1, x2 = _mk(1, 64, 2, 16, 16, seed=93), _mk(1, 32, 2, 16, 16, seed=94)
    w = _mk(96, 64, 3, 4, 4, seed=95) / (96 * 12) ** 0.5
    b = _mk(64, seed=96)
    aw, ab = _mk(64, 64, 1, 1, 1, seed=97) / 8.0, _mk(64, seed=98)

    def ref(a, c, w, b, aw, ab):
        v = F.conv_transpose3d(torch.cat([a, c], 1), w, b, (1, 2, 2), (1, 1, 1))
        return F.leaky_relu(v * torch.sigmoid(F.conv3d(v.mean((2, 3, 4), keepdim=True), aw, ab)), 0.2)

    _run(lambda a, c, w, b, aw, ab: ops.fused_conv3d(a, w, b, (1, 2, 2), (1, 1, 1), x2=c, transposed=True, se=(aw, ab),
                                                     act=ops.ACT_LRELU, slope=0.2),
         ref, [x1, x2, w, b, aw, ab], [True, True, True, True, True, True])


def test_virtual_concat_conv():
    x1, x2 = _mk(1, 64, 4, 10, 12, seed=7), _mk(1, 128, 4, 10, 12, seed=8)
    w = _mk(64, 192, 3, 3, 3, seed=9) / 72.0
    _run(lambda a, b, w: ops.fused_conv3d(a, w, None, 1, 1, x2=b),
         lambda a, b, w: F.conv3d(torch.cat([a, b], 1), w, None, 1, 1), [x1, x2, w], [True, True, True])


def test_virtual_concat_transposed():
    x1, x2 = _mk(1, 64, 3, 6, 7, seed=10), _mk(1, 64, 3, 6, 7, seed=11)
    w = _mk(128, 64, 3, 4, 4, seed=12) / 70.0
    _run(lambda a, b, w: ops.fused_conv3d(a, w, None, (1, 2, 2), 1, x2=b, transposed=True),
         lambda a, b, w: F.conv_transpose3d(torch.cat([a, b], 1), w, None, (1, 2, 2), 1), [x1, x2, w],
         [True, True, True])


def _se(v, aw, ab):
    return v * torch.sigmoid(F.conv3d(v.mean((2, 3, 4), keepdim=True), aw, ab))


@pytest.mark.parametrize("with_res", [True, False])
def test_conv_se_block(with_res):
    x = _mk(2, 64, 4, 9, 10, seed=13)
    w = _mk(64, 64, 3, 3, 3, seed=14) / 41.0
    b = _mk(64, seed=15)
    aw = _mk(64, 64, 1, 1, 1, seed=16) / 8.0
    ab = _mk(64, seed=17)
    res = _mk(2, 64, 4, 9, 10, seed=18)
    if with_res:
        _run(lambda x, w, b, aw, ab, r: ops.fused_conv3d(x, w, b, 1, 1, se=(aw, ab), res=r, act=ops.ACT_RELU),
             lambda x, w, b, aw, ab, r: torch.relu(_se(F.conv3d(x, w, b, 1, 1), aw, ab) + r),
             [x, w, b, aw, ab, res], [True] * 6)
    else:
        _run(lambda x, w, b, aw, ab: ops.fused_conv3d(x, w, b, 1, 1, se=(aw, ab), act=ops.ACT_LRELU, slope=0.2),
             lambda x, w, b, aw, ab: F.leaky_relu(_se(F.conv3d(x, w, b, 1, 1), aw, ab), 0.2),
             [x, w, b, aw, ab], [True] * 5)


def test_conv_instnorm_lrelu_block():
    x = _mk(2, 32, 5, 9, 11, seed=19)
    w = _mk(64, 32, 3, 3, 3, seed=20) / 29.0
    b = _mk(64, seed=21)
    ga = _mk(64, seed=22)
    be = _mk(64, seed=23)
    _run(lambda x, w, b, ga, be: ops.fused_conv3d(x, w, b, (1, 2, 2), 1, inorm=(ga, be), act=ops.ACT_LRELU,
                                                   slope=0.01),
         lambda x, w, b, ga, be: F.leaky_relu(
             F.instance_norm(F.conv3d(x, w, b, (1, 2, 2), 1), weight=ga, bias=be, eps=1e-5), 0.01),
         [x, w, b, ga, be], [True, True, False, True, True])  # d(bias) is identically 0 behind InstanceNorm


@pytest.mark.parametrize("Cin,Cout,K,stride,pad", [(1, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3)),
                                                    (2, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3)),
                                                    (1, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
                                                    (1, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1))])
def test_thin_input_conv(Cin, Cout, K, stride, pad):
    x = _mk(2, Cin, 5, 17, 19, seed=24)
    w = _mk(Cout, Cin, *K, seed=25) / (Cin * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(Cout, seed=26)
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, stride, pad, act=ops.ACT_RELU),
         lambda x, w, b: torch.relu(F.conv3d(x, w, b, stride, pad)), [x, w, b], [False, True, True])


@pytest.mark.parametrize("Cin,Cout,K,stride,pad,dims", [(1, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 6, 37, 150)),    # 3 column tiles, row tail
                                                         (2, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 3, 9, 70)),
                                                         (1, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 4, 12, 65)),
                                                         (1, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (3, 2, 8, 33)),
                                                         (1, 32, (3, 7, 7), (1, 2, 2), (1, 3, 3), (1, 4, 20, 140)),
                                                         (2, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), (1, 9, 64, 130))])
def test_thin_input_weight_gradient_kernel(Cin, Cout, K, stride, pad, dims):
    """thin_cin_wgrad_kernel (dY and x read once, voxel-reduction on v_mfma_f32_16x16x4_f32, bias column) against
    fp64 autograd: several tiles per block, column tiles, ragged rows / columns, both input-channel counts, every
    instantiated (C_out, tap-tile) shape; bitwise reproducible (slabs summed in a fixed order)."""
    from rehrseg_amd import hip_backend as hb
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=124).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    w = (_mk(Cout, Cin, *K, seed=125) / (Cin * K[0] * K[1] * K[2]) ** 0.5).to(_dev())
    cfg = ops.ConvCfg(stride, pad, False)
    od = tuple((i + 2 * p - k) // s + 1 for i, k, s, p in zip((D, H, W), K, stride, pad))
    dy = _mk(N, Cout, *od, seed=126).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    assert hb.small_cin_wgrad_on_mfma(x, w, dy, stride, pad)
    dw, db = ops.conv_wgrad(dy, x, None, w, cfg, True)
    dw2, db2 = ops.conv_wgrad(dy, x, None, w, cfg, True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    xr, wr = x.double().cpu(), w.double().cpu().requires_grad_()
    br = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    rw, rb = torch.autograd.grad(F.conv3d(xr, wr, br, stride, pad), [wr, br], dy.double().cpu())
    _close(dw, rw, 2e-5)
    _close(db, rb, 2e-5)


def test_thin_input_conv_instnorm():
    x = _mk(2, 1, 6, 12, 13, seed=27)
    w = _mk(32, 1, 3, 3, 3, seed=28) / 5.0
    b, ga, be = _mk(32, seed=29), _mk(32, seed=30), _mk(32, seed=31)
    _run(lambda x, w, b, ga, be: ops.fused_conv3d(x, w, b, 1, 1, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01),
         lambda x, w, b, ga, be: F.leaky_relu(F.instance_norm(F.conv3d(x, w, b, 1, 1), weight=ga, bias=be), 0.01),
         [x, w, b, ga, be], [False, True, False, True, True])


@pytest.mark.parametrize("Cin,Cout,K,pad,dims", [(32, 2, (1, 1, 1), (0, 0, 0), (2, 5, 9, 13)),
                                                 (16, 2, (5, 5, 5), (2, 2, 2), (2, 6, 9, 13)),
                                                 (32, 16, (3, 3, 3), (1, 1, 1), (1, 5, 9, 12)),
                                                 (64, 4, (1, 7, 7), (0, 3, 3), (1, 1, 20, 22))])
def test_thin_output_and_half_chunk_convs(Cin, Cout, K, pad, dims):
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=40)
    w = _mk(Cout, Cin, *K, seed=41) / (Cin * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(Cout, seed=42)
    _run(lambda x, w, b: ops.fused_conv3d(x, w, b, 1, pad), lambda x, w, b: F.conv3d(x, w, b, 1, pad),
         [x, w, b], [True, True, True])


@pytest.mark.parametrize("Di,scale", [(5, 4), (7, 2), (1, 3)])
def test_upsample_depth(Di, scale):
    x = _mk(2, 32, Di, 6, 7, seed=32)
    _run(lambda x: ops.upsample_depth(x, scale),
         lambda x: F.interpolate(x, scale_factor=(scale, 1, 1), mode="trilinear", align_corners=True), [x], [True])


@pytest.mark.parametrize("Di,scale,K", [(5, 4, (3, 3, 3)), (7, 2, (3, 3, 3)), (1, 3, (3, 3, 3)), (6, 4, (5, 3, 3))])
def test_upsample_conv3d_depth(Di, scale, K):
    """sr_head.0 on the depth-upsampled features (ref models/seg_model.py:204-205) without the upsampled tensor:
    (1,3,3) conv on the low-resolution slices + interpolate-and-sum-the-depth-taps, against interpolate -> conv -> ReLU."""
    x = _mk(2, 32, Di, 10, 12, seed=33)
    w = _mk(16, 32, *K, seed=34) / (32 * K[0] * K[1] * K[2]) ** 0.5
    b = _mk(16, seed=35)
    pad = tuple(k // 2 for k in K)
    _run(lambda x, w, b: ops.upsample_conv3d_depth(x, w, b, scale, act=ops.ACT_RELU),
         lambda x, w, b: F.relu(F.conv3d(F.interpolate(x, scale_factor=(scale, 1, 1), mode="trilinear",
                                                       align_corners=True), w, b, 1, pad)),
         [x, w, b], [True, True, True])


def test_rejects_cpu_tensors():
    from rehrseg_amd.lib import RehrsegHipError
    with pytest.raises(RehrsegHipError):
        ops.fused_conv3d(torch.randn(1, 32, 2, 4, 4), torch.randn(32, 32, 3, 3, 3), None, 1, 1)


def test_quad_maxpool_structure_loss():
    """Distiller's per-slice (H/2, W/2) max-pool (ref models/seg_model.py:95-113) on the device against nn.MaxPool2d,
    values and gradient (first maximum of a window)."""
    from rehrseg_amd.models.seg_model import _QuadMaxPool
    dev = _dev()
    x = _mk(2, 64, 3, 12, 10, seed=90)
    xg = x.to(dev).requires_grad_(True)
    y = _QuadMaxPool.apply(xg)
    xr = x.double().requires_grad_(True)
    fr = xr.permute(0, 2, 1, 3, 4).reshape(6, 64, 12, 10)
    ref = torch.nn.MaxPool2d(kernel_size=(6, 5), stride=(6, 5), padding=0, ceil_mode=True)(fr)
    _close(y, ref, 0.0)
    g = _mk(*ref.shape, seed=91)
    (gx,) = torch.autograd.grad(y, xg, g.to(dev))
    (rx,) = torch.autograd.grad(ref, xr, g.double())
    _close(gx, rx, 0.0)


def test_cosine_distance_loss_fused():
    """Distiller's cosine_distance_loss (ref models/seg_model.py:60-78) in two device passes against the torch
    composition in fp64: value and gradient w.r.t. the student tensor."""
    from rehrseg_amd.models import seg_model as sm
    dev = _dev()
    a, b = _mk(2, 64, 3, 10, 12, seed=92), _mk(2, 64, 3, 10, 12, seed=93) * 0.5 + 0.1
    ag = a.to(dev).requires_grad_(True)
    loss = sm.cosine_distance_loss(ag, b.to(dev))
    ar = a.double().requires_grad_(True)
    t1 = F.normalize(ar, p=2, dim=1).reshape(2, 64, -1)
    t2 = F.normalize(b.double(), p=2, dim=1).reshape(2, 64, -1)
    ref = (1 - torch.cosine_similarity(t1, t2, dim=2)).mean()
    assert abs(loss.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
    (g,) = torch.autograd.grad(loss, ag)
    (gr,) = torch.autograd.grad(ref, ar)
    _close(g, gr, 1e-4)


# ----------------------------------------------------------------------------- kernel-organisation variants
@pytest.mark.parametrize("Cin,Cout,dims", [(64, 64, (5, 40, 48)), (96, 128, (3, 32, 32)), (64, 192, (4, 32, 48))])
def test_winograd_big_tile_kernel_against_direct_kernel_and_fp64(Cin, Cout, dims):
    """wino_conv_big8_kernel (64 tiles x 64 channels, two waves per SIMD) against the direct gather-GEMM kernel and
    fp64, statistics epilogue included.  (Rounds 1-2 also carried a four-wave organisation of the same tile; it was
    removed in round 3 after the A/B of profiles/r03_ab_superseded.txt.)"""
    from rehrseg_amd import hip_backend as hb
    x = _mk(2, Cin, *dims, seed=61).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    w = (_mk(Cout, Cin, 3, 3, 3, seed=62) / (27 * Cin) ** 0.5).to(_dev())
    b = _mk(Cout, seed=63).to(_dev())
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    saved = hb.USE_WINOGRAD
    try:
        out = {}
        for flag in (False, True):
            hb.USE_WINOGRAD = flag
            before = hb.wino_launches
            out[flag] = ops.conv_forward(x, None, w, b, cfg, ops.ACT_LRELU, 0.01, 2)
            assert (hb.wino_launches > before) == flag
    finally:
        hb.USE_WINOGRAD = saved
    ref = F.leaky_relu(F.conv3d(x.double().cpu(), w.double().cpu(), b.double().cpu(), 1, 1), 0.01)
    _close(out[True][0], ref)
    _close(out[True][0], out[False][0].double().cpu(), 2e-5)
    torch.testing.assert_close(out[True][1], out[False][1], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dims,Cin,Cout", [((2, 4, 64, 48), 32, 32), ((1, 3, 60, 30), 48, 96), ((1, 2, 33, 50), 48, 32)])
def test_winograd_w32_pipelined_kernel(dims, Cin, Cout, monkeypatch):
    """wino_conv_w32p_kernel (32-channel tiles, 8 waves, software pipeline): its two block shapes (two 256-thread blocks
    per CU / one 512-thread block) do the same products in the same order per accumulator -> the same bits, statistics
    included; both against fp64; ragged regions and a half last chunk."""
    from rehrseg_amd import hip_backend as hb
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=164).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    w = (_mk(Cout, Cin, 3, 3, 3, seed=165) / (27 * Cin) ** 0.5).to(_dev())
    b = _mk(Cout, seed=166).to(_dev())
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    out = {}
    for blocks in (1, 2):
        monkeypatch.setattr(hb, "W32P_BLOCKS", blocks)
        out[blocks] = ops.conv_forward(x, None, w, b, cfg, ops.ACT_LRELU, 0.01, 2)
    assert torch.equal(out[1][0], out[2][0])
    torch.testing.assert_close(out[1][1], out[2][1], rtol=1e-6, atol=1e-6)
    ref = F.leaky_relu(F.conv3d(x.double().cpu(), w.double().cpu(), b.double().cpu(), 1, 1), 0.01)
    _close(out[1][0], ref)


@pytest.mark.parametrize("dims,Ca,Cg,K", [((16, 4, 12, 12), 128, 64, (3, 3, 3)),   # side-by-side mode, depth-major groups
                                          ((2, 4, 24, 32), 64, 64, (3, 3, 3)),     # plain lattice, od range per tap
                                          ((3, 2, 16, 16), 64, 128, (3, 3, 3)),    # depth 2: the outer taps see one slice
                                          ((20, 4, 12, 12), 64, 64, (3, 3, 3))])   # N % 8 != 0: no skipping side by side
def test_winograd_wgrad_tap_skip(dims, Ca, Cg, K):
    """Depth taps that skip the output slices without a source slice (REHR_WGRAD_NO_TAP_SKIP off) against the full walk:
    the skipped products are zeros, so weight and bias gradients agree to summation order; and against fp64."""
    from rehrseg_amd import hip_backend as hb
    N, D, H, W = dims
    x = _mk(N, Cg, D, H, W, seed=171).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    dy = _mk(N, Ca, D, H, W, seed=172).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    w = _mk(Ca, Cg, *K, seed=173).to(_dev())
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    saved = hb.USE_WGRAD_TAP_SKIP
    try:
        out = {}
        for flag in (False, True):
            hb.USE_WGRAD_TAP_SKIP = flag
            before = hb.wino_wgrad_launches
            out[flag] = ops.conv_wgrad(dy, x, None, w, cfg, True)
            assert hb.wino_wgrad_launches - before == 1
    finally:
        hb.USE_WGRAD_TAP_SKIP = saved
    _close(out[True][0], out[False][0].double().cpu(), 1e-5)   # (of the tensor's max: only the summation order differs)
    _close(out[True][1], out[False][1].double().cpu(), 1e-5)
    xr, wr = x.double().cpu().requires_grad_(), w.double().cpu().requires_grad_()
    br = torch.zeros(Ca, dtype=torch.float64, requires_grad=True)
    rw, rb = torch.autograd.grad(F.conv3d(xr, wr, br, 1, 1), [wr, br], dy.double().cpu())
    _close(out[True][0], rw)
    _close(out[True][1], rb)


def test_winograd_wgrad_64x64_block_against_direct_kernel_and_fp64():
    """The 64 x 64 block of wino_wgrad_kernel (two wave sets, two waves per SIMD) against the direct slab kernel and fp64."""
    from rehrseg_amd import hip_backend as hb
    x = _mk(2, 64, 4, 32, 48, seed=71).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    dy = _mk(2, 128, 4, 32, 48, seed=72).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    w = _mk(128, 64, 3, 3, 3, seed=73).to(_dev())
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    saved = hb.USE_WINOGRAD_WGRAD
    try:
        out = {}
        for flag in (False, True):
            hb.USE_WINOGRAD_WGRAD = flag
            before = hb.wino_wgrad_launches
            out[flag] = ops.conv_wgrad(dy, x, None, w, cfg, True)
            assert (hb.wino_wgrad_launches - before == 1) == flag
    finally:
        hb.USE_WINOGRAD_WGRAD = saved
    _close(out[True][0], out[False][0].double().cpu(), 1e-5)
    _close(out[True][1], out[False][1].double().cpu(), 1e-5)
    xr, wr = x.double().cpu().requires_grad_(), w.double().cpu().requires_grad_()
    br = torch.zeros(128, dtype=torch.float64, requires_grad=True)
    rw, rb = torch.autograd.grad(F.conv3d(xr, wr, br, 1, 1), [wr, br], dy.double().cpu())
    _close(out[True][0], rw)
    _close(out[True][1], rb)


@pytest.mark.parametrize("K,D,N,hw", [(16, 4, 2, (24, 40)), (4, 3, 1, (7, 9)), (8, 8, 3, (5, 5)), (32, 2, 2, (33, 17))])
def test_uasr_mix(K, D, N, hw):
    """rehr_uasr_mix_{fwd,bwd}_f32 against the reference's candidate loop in fp64 (oracle.flavr_oracle.uasr_head,
    FLAVR_arch.py:203-246): blended image / segmentation, uncertainty, and the gradients of both 1x1 responses and of
    uncertainty_out's weight and bias.  fp32 transcendental + summation-order differences only: 1e-5 of the scale."""
    from oracle.flavr_oracle import uasr_head
    om = _mk(N, D * 2 * K, 1, *hw, seed=81)
    ue = 2 * _mk(N, D * K, 1, *hw, seed=82)
    wu, bu = _mk(1, K, 1, 1, 1, seed=83), _mk(1, seed=84)
    dev = _dev()
    gin = [t.to(dev).requires_grad_() for t in (om, ue, wu, bu)]
    rin = [t.double().requires_grad_() for t in (om, ue, wu, bu)]
    assert ops.uasr_mix_supported(gin[0], gin[1], D)
    out, unc = ops.uasr_mix(*gin, D)
    ro, ru = uasr_head(rin[0][:, :, 0], rin[1][:, :, 0], rin[2], rin[3], D)
    _close(out, ro, 1e-5)
    _close(unc, ru, 1e-5)
    g0, g1 = _mk(*ro.shape, seed=85), _mk(*ru.shape, seed=86)
    got = torch.autograd.grad([out, unc], gin, [g0.to(dev), g1.to(dev)])
    ref = torch.autograd.grad([ro, ru], rin, [g0.double(), g1.double()])
    for a, e in zip(got, ref):
        _close(a.reshape(e.shape), e, 1e-5)


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("Cin,Cout,stride,dims", [(64, 32, (2, 2, 2), (2, 5, 12, 20)),      # 19 lattice tiles: padding blocks
                                                   (320, 256, (2, 2, 2), (1, 4, 4, 4)),      # several N tiles, one M tile
                                                   (128, 64, (1, 2, 2), (1, 6, 16, 16))])
def test_phase_interleaved_grid_is_bit_identical(Cin, Cout, stride, dims, mixed, monkeypatch):
    """Kernel = stride transposed convolution (nnU-Net UNetDecoder.transpconvs): the phases of a lattice tile as
    consecutive blocks of one XCD (REHR_DBG_GG_INTERLEAVE; measured slower, off by default) against blockIdx.z = phase -- the same blocks
    doing the same arithmetic in another order: identical bits, forward and the strided-conv input gradient alike; and
    against fp64."""
    from rehrseg_amd import hip_backend as hb
    N, D, H, W = dims
    dt = torch.bfloat16 if mixed else torch.float32
    x = _mk(N, Cin, D, H, W, seed=201).to(_dev()).to(dt).contiguous(memory_format=torch.channels_last_3d)
    w = (_mk(Cin, Cout, *stride, seed=202) / Cin ** 0.5).to(_dev())
    b = _mk(Cout, seed=203).to(_dev())
    cfg = ops.ConvCfg(stride, (0, 0, 0), True)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(hb, "PHASE_INTERLEAVE", flag)
        out[flag] = ops.conv_forward(x, None, w, b, cfg, ops.ACT_NONE, 0.0, 0)[0]
    assert torch.equal(out[True], out[False])
    wq = w.to(dt).double().cpu() if mixed else w.double().cpu()
    ref = F.conv_transpose3d(x.double().cpu(), wq, b.double().cpu(), stride)
    _close(out[True].float(), ref, 1e-2 if mixed else TOL)
    # the input gradient of the matching strided convolution walks the same phases
    w2 = (_mk(Cout, Cin, 3, 3, 3, seed=204) / (27 * Cin) ** 0.5).to(_dev())
    cfg2 = ops.ConvCfg(stride, (1, 1, 1), False)
    od = tuple((i + 2 - 3) // s + 1 for i, s in zip((D, H, W), stride))
    dz = _mk(N, Cout, *od, seed=205).to(_dev()).to(dt).contiguous(memory_format=torch.channels_last_3d)
    g = {}
    for flag in (True, False):
        monkeypatch.setattr(hb, "PHASE_INTERLEAVE", flag)
        g[flag] = ops.conv_dgrad(dz, w2, (D, H, W), Cin, 0, cfg2)[0]
    assert torch.equal(g[True], g[False])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("A,B,K", [(64, 64, (3, 3, 3)), (80, 48, (3, 3, 3)), (33, 20, (1, 3, 3)), (320, 256, (2, 2, 2)),
                                   (64, 64, (37, 3, 3)), (2, 16, (5, 5, 5)), (130, 7, (1, 1, 1))])
def test_pack_weights(A, B, K, dtype):
    """rehr_pack_weights_{f32,bf16} against a torch permute: both source layouts, padded rows zero."""
    from rehrseg_amd import hip_backend as hb
    T = K[0] * K[1] * K[2]
    Apad = ops.pad_rows(A)
    for transpose in (False, True):
        shape = (B, A) + K if transpose else (A, B) + K
        w = _mk(*shape, seed=300 + A + B).to(_dev())
        got = hb.pack_weights(w, A, Apad, B, T, transpose, dtype=dtype)
        src = w.reshape(shape[0], shape[1], T)
        want = (src.permute(2, 1, 0) if transpose else src.permute(2, 0, 1)).to(dtype)      # [t][a][b]
        assert tuple(got.shape) == (T, Apad, B)
        assert torch.equal(got[:, :A], want)
        assert not got[:, A:].any()


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("Cin,Cout,stride,dims,bias", [(64, 32, (2, 2, 2), (2, 5, 12, 20), True),     # 2400 voxels: ragged last tile
                                                        (128, 64, (1, 2, 2), (1, 6, 16, 16), True),
                                                        (32, 48, (2, 2, 2), (1, 3, 7, 9), False),       # padded C_out rows, no bias
                                                        (96, 128, (2, 2, 2), (1, 4, 8, 8), True),
                                                        (128, 96, (2, 1, 2), (2, 3, 5, 6), True)])
def test_transposed_conv_kernel_equals_stride_fused_kernel(Cin, Cout, stride, dims, bias, mixed, monkeypatch):
    """tconv_ks.hip (input tile staged once, all stride phases in one block) against the generic one-block-per-(tile, phase)
    grid and against fp64 on the same operands: forward of nnU-Net's UNetDecoder.transpconvs (seg_model.py:35)."""
    from rehrseg_amd import hip_backend as hb
    N, D, H, W = dims
    dt = torch.bfloat16 if mixed else torch.float32
    x = _mk(N, Cin, D, H, W, seed=211).to(_dev()).to(dt).contiguous(memory_format=torch.channels_last_3d)
    w = (_mk(Cin, Cout, *stride, seed=212) / Cin ** 0.5).to(_dev())
    b = _mk(Cout, seed=213).to(_dev()) if bias else None
    cfg = ops.ConvCfg(stride, (0, 0, 0), True)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(hb, "USE_TCONV_KS", flag)
        out[flag] = ops.conv_forward(x, None, w, b, cfg, ops.ACT_NONE, 0.0, 0)[0]
    assert out[True].dtype == dt
    _close(out[True].float(), out[False].double().cpu(), 2.0 ** -7 if mixed else 1e-5)
    wq = w.to(dt).double().cpu() if mixed else w.double().cpu()
    ref = F.conv_transpose3d(x.double().cpu(), wq, b.double().cpu() if bias else None, stride)
    _close(out[True].float(), ref, 1e-2 if mixed else TOL)


@pytest.mark.parametrize("Cin,Cout,dims", [(32, 32, (2, 7, 48, 40)), (64, 64, (1, 5, 32, 48)), (32, 32, (1, 3, 64, 48))])
def test_winograd_tile_orders_are_bit_identical(Cin, Cout, dims, monkeypatch):
    """Band-major tile order (an XCD walks depth inside a band of rows: default) against slice-major
    (REHR_DBG_GG_SLICE_MAJOR): the same tiles in another order -> the same bits, statistics to summation order."""
    from rehrseg_amd import hip_backend as hb
    N, D, H, W = dims
    x = _mk(N, Cin, D, H, W, seed=221).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    w = (_mk(Cout, Cin, 3, 3, 3, seed=222) / (27 * Cin) ** 0.5).to(_dev())
    b = _mk(Cout, seed=223).to(_dev())
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(hb, "WINO_BAND_MAJOR", flag)
        out[flag] = ops.conv_forward(x, None, w, b, cfg, ops.ACT_LRELU, 0.01, 2)
    assert torch.equal(out[True][0], out[False][0])
    torch.testing.assert_close(out[True][1], out[False][1], rtol=1e-9, atol=1e-6)


def test_winograd_wgrad_tap_colocation_is_bit_identical(monkeypatch):
    """The depth taps of a split as consecutive blocks of one XCD (default where the splits come in whole groups of 8)
    against blockIdx.z = tap: the same blocks, the same slabs, the same fixed-order reduction."""
    from rehrseg_amd import hip_backend as hb
    x = _mk(2, 32, 24, 64, 64, seed=231).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    dy = _mk(2, 32, 24, 64, 64, seed=232).to(_dev()).contiguous(memory_format=torch.channels_last_3d)
    w = _mk(32, 32, 3, 3, 3, seed=233).to(_dev())
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(hb, "WGRAD_TAP_COLOCATE", flag)
        before = hb.wino_wgrad_launches
        out[flag] = ops.conv_wgrad(dy, x, None, w, cfg, True)
        assert hb.wino_wgrad_launches - before == 1
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    xr, wr = x.double().cpu(), w.double().cpu().requires_grad_()
    br = torch.zeros(32, dtype=torch.float64, requires_grad=True)
    rw, rb = torch.autograd.grad(F.conv3d(xr, wr, br, 1, 1), [wr, br], dy.double().cpu())
    _close(out[True][0], rw)
    _close(out[True][1], rb)
