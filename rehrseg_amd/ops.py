"""Host-side operators of the hot path: geometry (stride phases, tap tables,
lattice tiles), weight packing and the autograd glue around the C-ABI kernels.

What each operator replaces in the reference (all torch.nn eager ops there):
  fused_conv3d(mode="plain")  Conv3d(+ReLU)            models/FLAVR/resnet_3D.py:19-33,122-133,196-200
  fused_conv3d(se=...)        Conv3d/ConvTranspose3d -> SEGating (-> +residual) -> ReLU/LeakyReLU
                                                       resnet_3D.py:100-116,140-151; FLAVR_arch.py:40-88,188-200
  fused_conv3d(inorm=...)     Conv3d -> InstanceNorm3d -> LeakyReLU (nnU-Net ConvDropoutNormReLU,
                                                       dynamic_network_architectures==0.3.1; seg_model.py:174-191)
  fused_conv3d(transposed)    ConvTranspose3d          FLAVR_arch.py:50; UNetDecoder.transpconvs (seg_model.py:35)
  upsample_depth              F.interpolate(scale=(u,1,1), trilinear, align_corners=True)  seg_model.py:204

PyTorch supplies autograd bookkeeping, device memory and the stream; every
arithmetic step is a kernel of librehrseg_hip.so.
"""
from __future__ import annotations

import functools
import itertools
import weakref
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import hip_backend

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2

_backend = hip_backend


def get_backend():
    return _backend


def set_backend(b):
    """Tests swap in a CPU emulator of the C-ABI to exercise this host logic
    without a GPU; the product never calls this."""
    global _backend
    old = _backend
    _backend = b
    return old


# ----------------------------------------------------------------------------- mixed precision
# BASELINE.json configs[4]: "bf16 mixed precision ... MFMA implicit-GEMM path".  Inside `mixed_precision()` every
# matrix-core convolution (and the SEGating / InstanceNorm tail fused behind it) takes bf16 activations and a bf16
# copy of the fp32 master weights, accumulates in fp32, forms its statistics in fp32/fp64 and returns bf16
# activations; gradients of activations are bf16, weight gradients fp32.  The thin layers (1-2 input channels,
# <= 4 output channels), the depth upsample and the losses stay fp32 -- the same split torch.autocast makes.
_mixed = False


class mixed_precision:
    """Context manager: run the matrix-core convolutions in bf16 (fp32 accumulate, fp32 master weights)."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        global _mixed
        self.prev, _mixed = _mixed, self.enabled
        return self

    def __exit__(self, *exc):
        global _mixed
        _mixed = self.prev
        return False


def is_mixed_precision():
    return _mixed


# ----------------------------------------------------------------------------- layout
def to_cl(x: torch.Tensor) -> torch.Tensor:
    """Logical (N,C,D,H,W) tensor whose memory is NDHWC-dense; fp32, or bf16 on the mixed-precision path."""
    if x.dim() != 5:
        raise ValueError(f"expected a 5-D (N,C,D,H,W) tensor, got {tuple(x.shape)}")
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float64):  # float64 only reaches the CPU emulator in tests
        x = x.float()
    if x.permute(0, 2, 3, 4, 1).is_contiguous():
        return x
    return x.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)


def _triple(v):
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(a) for a in v)
    if len(v) != 3:
        raise ValueError(f"expected 3 values, got {v}")
    return v


# ----------------------------------------------------------------------------- geometry
def conv_out_dim(i, k, s, p):
    return (i + 2 * p - k) // s + 1


def phase_taps(K, s, p, phase):
    """Taps of a stride-``s`` / kernel-``K`` / pad-``p`` convolution that touch the
    positions ``== phase (mod s)`` of its un-strided side (input gradient of a
    Conv3d, output of a ConvTranspose3d).  Position ``s*q + phase`` reads the
    strided side at ``q + off0 - j`` with weight index ``k0 + s*j``."""
    k0 = (phase + p) % s
    if k0 >= K:
        return None
    count = (K - k0 + s - 1) // s
    off0 = (phase + p - k0) // s
    return (count, off0, -1, k0, s)


def full_taps(K):
    return (K, 0, 1, 0, 1)


@functools.lru_cache(maxsize=None)
def choose_tile(lattice, thin=(False, False, False)):
    """128-voxel lattice tile with the least padding (ties: smaller halo, wider w).
    Returns (0,0,0) = runs of 128 flattened points when bricks waste > 10 % more.
    An axis flagged `thin` (its source has extent 1 but several taps point at it, e.g.
    feature_fuse's input gradient) keeps tile extent 1 there: every row of a tile then
    uses the same single tap instead of multiplying zeros for the others."""
    Ld, Lh, Lw = lattice
    best = None
    for td in (1, 2, 4, 8, 16, 32, 64, 128):
        for th in (1, 2, 4, 8, 16, 32, 64, 128):
            if 128 % (td * th):
                continue
            tw = 128 // (td * th)
            if (thin[0] and td > 1) or (thin[1] and th > 1) or (thin[2] and tw > 1):
                continue
            vol = (-(-Ld // td) * td) * (-(-Lh // th) * th) * (-(-Lw // tw) * tw)
            halo = (td + 2) * (th + 2) * (tw + 2)
            key = (vol, halo, -tw)
            if best is None or key < best[0]:
                best = (key, (td, th, tw))
    lin = -(-(Ld * Lh * Lw) // 128) * 128
    if best is None or (best[0][0] > 1.10 * lin and not any(thin)):
        return (0, 0, 0)
    return best[1]


def pad_rows(a):
    """Row padding of a packed weight panel (gather-GEMM N tiles are 32/64/128)."""
    if a % 128 == 0 or a % 64 == 0:
        return a
    return -(-a // 32) * 32


@dataclass(frozen=True)
class ConvCfg:
    stride: Tuple[int, int, int]
    pad: Tuple[int, int, int]
    transposed: bool = False
    act: int = ACT_NONE
    slope: float = 0.0
    mode: str = "plain"  # plain | se | in
    eps: float = 1e-5
    y_fp32: bool = False  # mixed precision: keep a thin-input layer's output fp32 (linear post-processing follows)


def _spatial(x):
    return tuple(x.shape[2:])


def _out_dims(in_dims, w_shape, cfg: ConvCfg):
    K = tuple(w_shape[2:])
    if cfg.transposed:
        return tuple((i - 1) * s - 2 * p + k for i, k, s, p in zip(in_dims, K, cfg.stride, cfg.pad))
    return tuple(conv_out_dim(i, k, s, p) for i, k, s, p in zip(in_dims, K, cfg.stride, cfg.pad))


def _pack(w, dest_dim, dtype=torch.float32, part=None):
    """wp[tap][dest rows (padded)][other dim] in `dtype` (the activations' dtype: fp32, or bf16 on the mixed-precision
    path -- the master weights `w` stay fp32); dest_dim = which weight dim feeds the output channels.
    part = (dim, lo, count): of w.narrow(dim, lo, count) (one half of a virtual channel concat; the backend keeps the
    panels of parameters between steps, so it is told about the parameter, not handed a temporary slice)."""
    be = get_backend()
    T = w.shape[2] * w.shape[3] * w.shape[4]
    kw = {"dtype": torch.bfloat16} if dtype == torch.bfloat16 else {}
    shape = list(w.shape)
    if part is not None:
        if hasattr(be, "invalidate_weight_forms"):
            kw["part"] = part
        else:
            w = w.narrow(*part).contiguous()
        shape[part[0]] = part[2]
    if dest_dim == 0:
        A, B = shape[0], shape[1]
        return be.pack_weights(w, A, pad_rows(A), B, T, False, **kw), pad_rows(A)
    A, B = shape[1], shape[0]
    return be.pack_weights(w, A, pad_rows(A), B, T, True, **kw), pad_rows(A)


def _phased_gather(src1, src2, c1, Csrc, wp, Npad, dst, Cdst, K, stride, pad, bias, act, slope, stats, stats_mode):
    """dst positions of every stride phase <- taps of src (Conv3d input gradient /
    ConvTranspose3d forward)."""
    be = get_backend()
    src_dims, dst_dims = _spatial(src1), _spatial(dst)
    calls = []
    for ph in itertools.product(*(range(s) for s in stride)):
        taps = [phase_taps(K[a], stride[a], pad[a], ph[a]) for a in range(3)]
        lattice = tuple((dst_dims[a] - ph[a] + stride[a] - 1) // stride[a] for a in range(3))
        if min(lattice) <= 0:
            continue
        if any(t is None for t in taps):
            continue  # no tap reaches this phase: dst keeps its zero fill
        thin = tuple(src_dims[a] == 1 and taps[a][0] > 1 for a in range(3))
        calls.append((src1, src2, c1, src_dims, Csrc, lattice, (1, 1, 1), (0, 0, 0), taps, K[1], K[2], wp, Npad,
                      dst, dst_dims, Cdst, stride, ph, bias, act, slope, stats, stats_mode,
                      choose_tile(tuple(lattice), thin)))
    if calls:
        be.gather_gemm_multi(calls)  # all phases of the layer in one grid


def _has_empty_phase(K, stride, pad):
    return any(phase_taps(K[a], stride[a], pad[a], ph) is None for a in range(3) for ph in range(stride[a]))


def _thin_out(w, cfg: ConvCfg, has_x2):
    """Segmentation-logit convs (Cout <= 4, stride 1): direct HBM-bound kernels, not MFMA tiles."""
    return (not cfg.transposed and not has_x2 and w.shape[0] <= 4 and w.shape[1] % 16 == 0 and
            cfg.stride == (1, 1, 1) and w.shape[4] <= 7)


def _thin5_f32(be, x, w, cfg: ConvCfg, shape=None):
    """sr_head.2 (16 -> 2, 5x5x5) in fp32 on a shape the fp32 matrix-core kernels take (thin_conv_f32.hip)."""
    return (x.dtype == torch.float32 and x.is_cuda and hasattr(be, "thin5_supported") and
            be.thin5_supported(tuple(shape if shape is not None else x.shape), tuple(w.shape), cfg.pad, torch.float32))


WINO_TAP_SPLIT = True   # see _tap_split (tests and A/B runs switch it off)


def _sub_taps(t, a, b):
    """taps [a, b) of one axis' arithmetic tap description."""
    cnt, off0, offs, k0, ks = t
    return (b - a, off0 + offs * a, offs, k0 + ks * a, ks)


def _tap_split(lattice, N, Npad, taps, Cin, bf16=False):
    """Tap ranges (<= 8) for a split-K launch, or None when the plain launch already fills the chip.
    Depth taps are split first (single taps or ranges), then the row taps in two.
    bf16: no Winograd kernels to leave lattices to (the LDS halo-brick kernel declines these few-tile lattices or would
    run them on a handful of CUs: the split gather launch fills the chip)."""
    vox = lattice[0] * lattice[1] * lattice[2]
    blocks = -(-vox // 128) * (Npad // (128 if Npad % 128 == 0 else (64 if Npad % 64 == 0 else 32))) * N
    T = taps[0][0] * taps[1][0] * taps[2][0]
    if blocks > 160 or T * Cin < 2048 or taps[0][0] * taps[1][0] < 2:
        return None
    if not bf16 and taps[0][0] <= 3 and taps[1][0] == 3 and taps[2][0] == 3 and taps[1][2] in (1, -1) and taps[2][2] in (1, -1):
        # unit-stride 3x3 taps: leave lattices the Winograd kernels accept (8x16-output regions, <= 1.3x
        # padding; librehrseg's wino_workspace_bytes applies the same rule) to them
        Lh, Lw = lattice[1], lattice[2]
        if (WINO_TAP_SPLIT and taps[0][0] == 3 and lattice[0] >= 2 and Lh % 16 == 0 and Lw % 16 == 0 and Npad % 64 == 0 and
                (Lh // 16) * (Lw // 16) * lattice[0] * N * (Npad // 64) <= 160):
            # whole 16x16 regions but at most ~half a chip of big-tile Winograd blocks (nnU-Net's 16^3 stage: 128): the
            # three depth taps as three parts of ONE Winograd grid (wino_conv_split_try), a third of the K loop each
            return [[_sub_taps(taps[0], a, a + 1), taps[1], taps[2]] for a in range(3)]
        if Lh >= 8 and Lw >= 8 and (-(-Lh // 8) * 8) * (-(-Lw // 16) * 16) * 100 <= Lh * Lw * 134:
            return None
        # ... and small planes the flattened-tile Winograd kernel takes (16..63 tiles per slice, >= 64 tiles)
        tps = -(-Lh // 2) * -(-Lw // 2)
        if (6 <= Lh <= 16 and 6 <= Lw <= 16 and 16 <= tps < 64 and Npad % 64 == 0 and
                -(-(N * lattice[0] * tps) // 64) * (Npad // 64) >= 128):
            return None
    want = min(8, max(2, 512 // blocks))
    kd, kh = taps[0][0], taps[1][0]
    nd = min(kd, want)
    nh = 2 if (kh >= 2 and nd * 2 <= want) else 1
    dstep, hstep = -(-kd // nd), -(-kh // nh)
    parts = []
    for a in range(0, kd, dstep):
        for c in range(0, kh, hstep):
            parts.append([_sub_taps(taps[0], a, min(kd, a + dstep)), _sub_taps(taps[1], c, min(kh, c + hstep)), taps[2]])
    return parts if len(parts) > 1 else None


def conv_forward(x1, x2, w, bias, cfg: ConvCfg, act, slope, stats_mode):
    """Returns (y, stats).  x2 = second half of a virtual channel concat."""
    be = get_backend()
    N = x1.shape[0]
    c1 = x1.shape[1]
    Cin = c1 + (x2.shape[1] if x2 is not None else 0)
    K = tuple(w.shape[2:])
    in_dims = _spatial(x1)
    out_dims = _out_dims(in_dims, w.shape, cfg)
    Cout = w.shape[1] if cfg.transposed else w.shape[0]
    if (w.shape[0] if cfg.transposed else w.shape[1]) != Cin:
        raise ValueError(f"weight {tuple(w.shape)} does not match {Cin} input channels")
    stats = None
    if stats_mode:
        stats = (be.zeros_f64((N, Cout, 2), x1.device) if hasattr(be, "zeros_f64")
                 else torch.zeros((N, Cout, 2), dtype=torch.float64, device=x1.device))
    if Cin <= 2:
        if cfg.transposed or x2 is not None:
            raise ValueError("thin-input path handles plain Conv3d only")
        # mixed precision: fp32 arithmetic on the fp32 image, the result stored as bf16 for the layer behind (no cast
        # pass over the network's largest activation); statistics from the fp32 accumulators
        okw = {}
        if (_mixed and not cfg.y_fp32 and x1.dtype == torch.float32 and hasattr(be, "small_cin_bf16_out_ok") and
                be.small_cin_bf16_out_ok(tuple(w.shape), cfg.stride)):
            okw["dtype"] = torch.bfloat16
        y = be.new_act(N, Cout, *out_dims, like=x1, **okw)
        be.small_cin_fwd(x1, w, bias, y, cfg.stride, cfg.pad, act, slope, stats, stats_mode)
        return y, stats
    if _thin_out(w, cfg, x2 is not None):
        if stats_mode:
            raise NotImplementedError("statistics epilogue on a thin-output conv")
        if x1.dtype == torch.bfloat16:   # sr_head.2 on the matrix cores (fused_conv3d only routes supported shapes here)
            if act != ACT_NONE:
                raise NotImplementedError("activation behind the bf16 thin-output conv")
            return be.thin5_fwd(x1, w, bias), None
        if act == ACT_NONE and _thin5_f32(be, x1, w, cfg):
            return be.thin5_fwd(x1, w, bias), None
        y = be.new_act(N, Cout, *out_dims, like=x1)
        be.small_cout_fwd(x1, w, bias, y, cfg.pad, act, slope)
        return y, None
    if cfg.transposed:
        empty = _has_empty_phase(K, cfg.stride, cfg.pad)
        if empty and bias is not None:
            raise NotImplementedError("ConvTranspose3d with unreachable output phases and a bias")
        y = be.new_act(N, Cout, *out_dims, like=x1, zero=empty)
        wp, Npad = _pack(w, 1, x1.dtype)
        _phased_gather(x1, x2, c1, Cin, wp, Npad, y, Cout, K, cfg.stride, cfg.pad, bias, act, slope, stats,
                       stats_mode)
        return y, stats
    wp, Npad = _pack(w, 0, x1.dtype)
    taps = [full_taps(k) for k in K]
    # Few lattice tiles but a long K (feature_fuse: 128 tiles x 1152 taps; nnU-Net stages at <= 8^3 voxels:
    # 40 tiles x 27 taps x 320 channels): split the taps over S partial launches in one grid and combine the
    # slabs in a fixed order (the combine carries bias, activation and the statistics epilogue).
    bf16 = x1.dtype == torch.bfloat16   # (mixed precision: fp32 slabs from the bf16 kernels, bf16 out of the combine)
    # (strided convolutions too: nnU-Net's 8^3 -> 4^3 stage is 10 blocks walking 8640 products each without it)
    parts = _tap_split(out_dims, N, Npad, taps, Cin, bf16)
    if parts is not None and Cout % 4 == 0:
        S = len(parts)
        slabs = be.new_act(S * N, Cout, *out_dims, like=x1, **({"dtype": torch.float32} if bf16 else {}))
        calls = [(x1, x2, c1, in_dims, Cin, out_dims, cfg.stride, tuple(-p for p in cfg.pad), tp, K[1], K[2],
                  wp, Npad, slabs[s_i * N:(s_i + 1) * N], out_dims, Cout, (1, 1, 1), (0, 0, 0), None,
                  ACT_NONE, 0.0, None, 0, choose_tile(tuple(out_dims))) for s_i, tp in enumerate(parts)]
        be.gather_gemm_multi(calls)
        okw = {"out_dtype": torch.bfloat16} if bf16 else {}
        return be.sum_slabs_bias_act(slabs, S, bias, act, slope, stats if stats_mode else None, **okw), stats
    y = be.new_act(N, Cout, *out_dims, like=x1)
    be.gather_gemm(x1, x2, c1, in_dims, Cin, out_dims, cfg.stride, tuple(-p for p in cfg.pad), taps, K[1], K[2],
                   wp, Npad, y, out_dims, Cout, (1, 1, 1), (0, 0, 0), bias, act, slope, stats, stats_mode,
                   choose_tile(tuple(out_dims)))
    return y, stats


def conv_dgrad(dz, w, in_dims, c1, c2, cfg: ConvCfg, need1=True, need2=True, x_dtype=None):
    """Input gradient(s) of conv_forward: (dx1, dx2).  `x_dtype`: dtype of the layer's input where it differs from
    dz's (the bf16 thin-output head takes bf16 in, fp32 out)."""
    be = get_backend()
    N = dz.shape[0]
    K = tuple(w.shape[2:])
    Cz = dz.shape[1]
    if _thin_out(w, cfg, c2 > 0):
        if x_dtype == torch.bfloat16:
            return (be.thin5_dgrad(dz.float() if dz.dtype != torch.float32 else dz, w) if need1 else None), None
        if dz.dtype == torch.float32 and _thin5_f32(be, dz, w, cfg, (N, c1) + tuple(in_dims)):
            return (be.thin5_dgrad(dz, w, torch.float32) if need1 else None), None
        return (be.small_cout_dgrad(dz, w, (N, c1) + tuple(in_dims), cfg.pad) if need1 else None), None
    out = []
    for (lo, cnt, need) in ((0, c1, need1), (c1, c2, need2)):
        if cnt == 0 or not need:
            out.append(None)
            continue
        if cfg.transposed:
            # dX of ConvTranspose3d = strided Conv3d of dY with w[ci][co] (dest rows = dim 0)
            wp, Npad = _pack(w, 0, dz.dtype, None if (lo == 0 and cnt == w.shape[0]) else (0, lo, cnt))
            dx = be.new_act(N, cnt, *in_dims, like=dz)
            taps = [full_taps(k) for k in K]
            be.gather_gemm(dz, None, Cz, _spatial(dz), Cz, in_dims, cfg.stride, tuple(-p for p in cfg.pad), taps,
                           K[1], K[2], wp, Npad, dx, in_dims, cnt, (1, 1, 1), (0, 0, 0), None, ACT_NONE, 0.0, None,
                           0, choose_tile(tuple(in_dims)))
        else:
            wp, Npad = _pack(w, 1, dz.dtype, None if (lo == 0 and cnt == w.shape[1]) else (1, lo, cnt))
            if cfg.stride == (1, 1, 1) and cnt % 4 == 0:
                # low-resolution stages: split the taps like the forward does (one phase, no epilogue)
                taps = [phase_taps(K[a], 1, cfg.pad[a], 0) for a in range(3)]
                zbf = dz.dtype == torch.bfloat16
                parts = _tap_split(in_dims, N, Npad, taps, Cz, zbf) if all(t is not None for t in taps) else None
                if parts is not None:
                    S = len(parts)
                    slabs = be.new_act(S * N, cnt, *in_dims, like=dz, **({"dtype": torch.float32} if zbf else {}))
                    be.gather_gemm_multi([(dz, None, Cz, _spatial(dz), Cz, tuple(in_dims), (1, 1, 1), (0, 0, 0), tp, K[1],
                                           K[2], wp, Npad, slabs[s_i * N:(s_i + 1) * N], tuple(in_dims), cnt, (1, 1, 1),
                                           (0, 0, 0), None, ACT_NONE, 0.0, None, 0, choose_tile(tuple(in_dims)))
                                          for s_i, tp in enumerate(parts)])
                    out.append(be.sum_slabs_bias_act(slabs, S, None, ACT_NONE, 0.0,
                                                     **({"out_dtype": torch.bfloat16} if zbf else {})))
                    continue
            dx = be.new_act(N, cnt, *in_dims, like=dz, zero=_has_empty_phase(K, cfg.stride, cfg.pad))
            _phased_gather(dz, None, Cz, Cz, wp, Npad, dx, cnt, K, cfg.stride, cfg.pad, None, ACT_NONE, 0.0, None, 0)
        out.append(dx)
    return out[0], out[1]


# Parameters whose .grad is a view into a caller-owned flat buffer (rehrseg_amd.parallel.PatchParallel):
# weight.data_ptr() -> (weakref(parameter), weakref(owner)).  The weight-gradient kernels then write straight
# into that view and the owner is told the gradient is ready, instead of returning a temporary for autograd to
# add into it (one extra elementwise pass and one allocation per parameter and step).  Weak references: a
# registry entry never keeps a model alive, and an address reused by another tensor does not match a dead one.
_direct_grad = {}


def register_direct_grad(param, owner):
    _direct_grad[param.data_ptr()] = (weakref.ref(param), weakref.ref(owner))


def unregister_direct_grad(owner):
    for k in [k for k, (_, o) in _direct_grad.items() if o() is owner or o() is None]:
        del _direct_grad[k]


def _direct_entry(w):
    ent = _direct_grad.get(w.data_ptr())
    if ent is None:
        return None
    param, owner = ent[0](), ent[1]()
    if param is None or owner is None or param.data_ptr() != w.data_ptr() or param.shape != w.shape:
        del _direct_grad[w.data_ptr()]
        return None
    return param, owner


def _note_use(w):
    """Forward of a node that will want this weight's gradient: the owner counts the uses per step, a weight
    used more than once keeps autograd's accumulation (a direct write would be overwritten by the next use)."""
    ent = _direct_entry(w)
    if ent is not None:
        ent[1].note_use(ent[0])


def _direct_grad_target(w):
    ent = _direct_entry(w)
    if ent is None:
        return None
    param, owner = ent
    g = param.grad
    if g is None or g.shape != w.shape or not g.is_contiguous() or not owner.may_write(param):
        return None
    return param, owner, g


def conv_wgrad(dz, x1, x2, w, cfg: ConvCfg, want_bias, out=None):
    """(dw, db) in the torch parameter layout of ``w``."""
    be = get_backend()
    N = x1.shape[0]
    K = tuple(w.shape[2:])
    T = K[0] * K[1] * K[2]
    c1 = x1.shape[1]
    Cin = c1 + (x2.shape[1] if x2 is not None else 0)
    if Cin <= 2:
        # thin input: im2col columns (k = ci*T + tap) + a 1x1x1 weight gradient on the MFMA path
        Cout, kcols = w.shape[0], Cin * T
        if Cout % 32 or (hasattr(be, "small_cin_wgrad_on_mfma") and
                         be.small_cin_wgrad_on_mfma(x1, w, dz, cfg.stride, cfg.pad)):
            return be.small_cin_wgrad(x1, w, dz, cfg.stride, cfg.pad, want_bias)   # dY and x read once each
        if dz.dtype == torch.bfloat16:
            dz = dz.float()
        kpad = -(-kcols // 32) * 32
        col = be.im2col(x1, w, _spatial(dz), cfg.stride, cfg.pad, kpad)
        tmp = torch.empty((Cout, kpad), dtype=w.dtype, device=w.device)
        db = torch.empty((Cout,), dtype=w.dtype, device=w.device) if want_bias else None
        one = [full_taps(1)] * 3
        be.wgrad(dz, Cout, col, kpad, N, _spatial(dz), _spatial(dz), (1, 1, 1), (0, 0, 0), one, 1, 1, tmp, 0,
                 (kpad, 1, 0), False, db)
        return tmp[:, :kcols].reshape(w.shape).contiguous(), db
    if _thin_out(w, cfg, x2 is not None):
        if x1.dtype == torch.bfloat16:
            return be.thin5_wgrad(x1, w, dz.float() if dz.dtype != torch.float32 else dz, want_bias)
        if dz.dtype == torch.float32 and _thin5_f32(be, x1, w, cfg):
            return be.thin5_wgrad(x1, w, dz, want_bias)
        return be.small_cout_wgrad(x1, w, dz, cfg.pad, want_bias)
    dw = out if out is not None else torch.empty(tuple(w.shape), dtype=w.dtype, device=w.device)
    db = None
    taps = [full_taps(k) for k in K]
    b = tuple(-p for p in cfg.pad)
    fused_bias = dz.dtype != torch.bfloat16   # the mixed-precision kernel leaves the bias gradient to a column sum
    if cfg.transposed:
        # lattice = input x (rows of w), gathered = dY (cols of w)
        Cout = w.shape[1]
        lo = 0
        for xs in (x1, x2):
            if xs is None:
                continue
            cnt = xs.shape[1]
            be.wgrad(xs, cnt, dz, Cout, N, _spatial(xs), _spatial(dz), cfg.stride, b, taps, K[1], K[2], dw,
                     lo * Cout * T, (Cout * T, T, 1), False, None)
            lo += cnt
        if want_bias:
            db = be.channel_sum(dz)
    else:
        Cout = w.shape[0]
        db = torch.empty((Cout,), dtype=w.dtype, device=w.device) if (want_bias and fused_bias) else None
        lo = 0
        first = True
        for xs in (x1, x2):
            if xs is None:
                continue
            cnt = xs.shape[1]
            be.wgrad(dz, Cout, xs, cnt, N, _spatial(dz), _spatial(xs), cfg.stride, b, taps, K[1], K[2], dw, lo * T,
                     (Cin * T, T, 1), False, db if first else None)
            first = False
            lo += cnt
        if want_bias and not fused_bias:
            db = be.channel_sum(dz)
    return dw, db


# ----------------------------------------------------------------------------- autograd
class _FusedConv(torch.autograd.Function):
    """conv / transposed conv, optionally followed by SEGating(+residual)+act or
    InstanceNorm+act, as one autograd node."""

    @staticmethod
    def forward(ctx, x1, x2, w, b, p1, p2, res, cfg: ConvCfg):
        be = get_backend()
        x1 = to_cl(x1)
        x2 = to_cl(x2) if x2 is not None else None
        res = to_cl(res) if res is not None else None
        N = x1.shape[0]
        if cfg.mode == "plain":
            y, _ = conv_forward(x1, x2, w, b, cfg, cfg.act, cfg.slope, 0)
            y0 = gate = mean = mr = None
        elif cfg.mode == "se":
            y0, stats = conv_forward(x1, x2, w, b, cfg, ACT_NONE, 0.0, 1)
            Cc = y0.shape[1]
            S = y0.shape[2] * y0.shape[3] * y0.shape[4]
            gate, mean = be.se_gate_fwd(stats, p1.reshape(Cc, Cc), p2, N, Cc, S)
            y = be.scale_res_act_fwd(y0, gate, res, cfg.act, cfg.slope)
            mr = None
        elif cfg.mode == "in":
            y0, stats = conv_forward(x1, x2, w, b, cfg, ACT_NONE, 0.0, 2)
            y, mr = be.instnorm_act_fwd(y0, stats, p1, p2, cfg.eps, cfg.act, cfg.slope)
            gate = mean = None
        else:
            raise ValueError(cfg.mode)
        ctx.cfg = cfg
        ctx.has = (x2 is not None, b is not None, res is not None)
        ctx.save_for_backward(x1, x2, w, p1, p2, y0, y, gate, mean, mr)
        ctx.bias_ref = weakref.ref(b) if isinstance(b, torch.nn.Parameter) else None
        if _direct_grad and ctx.needs_input_grad[2]:
            _note_use(w)
        return y

    @staticmethod
    def backward(ctx, dy):
        be = get_backend()
        be.keep_forms = not ctx.needs_input_grad[2]
        cfg = ctx.cfg
        x1, x2, w, p1, p2, y0, y, gate, mean, mr = ctx.saved_tensors
        has_x2, has_b, has_res = ctx.has
        dy = to_cl(dy)
        dp1 = dp2 = dres = None
        if cfg.mode == "plain":
            dz = be.act_bwd(dy, y, cfg.act, cfg.slope) if cfg.act != ACT_NONE else dy
        elif cfg.mode == "se":
            Cc = y0.shape[1]
            S = y0.shape[2] * y0.shape[3] * y0.shape[4]
            dz, dres, dgate = be.scale_res_act_bwd(dy, y, y0, gate, has_res, cfg.act, cfg.slope)
            dp1, dp2, k = be.se_gate_bwd(dgate, gate, mean, p1.reshape(Cc, Cc), S)
            dp1 = dp1.reshape(p1.shape)
            be.add_channel_const(dz, k)
        else:
            db_in = None
            if (has_b and ctx.needs_input_grad[3] and y0.dtype == torch.bfloat16 and not cfg.transposed and
                    hasattr(be, "small_cin_bf16_out_ok")):
                # mixed precision: the conv-bias gradient (column sums of dz) rides on the InstanceNorm apply pass
                dz, dp1, dp2, db_in = be.instnorm_act_bwd(dy, y0, mr, p1, p2, cfg.act, cfg.slope, want_conv_bias=True)
            else:
                dz, dp1, dp2 = be.instnorm_act_bwd(dy, y0, mr, p1, p2, cfg.act, cfg.slope)
        need = ctx.needs_input_grad
        c1 = x1.shape[1]
        c2 = x2.shape[1] if has_x2 else 0
        dx1 = dx2 = None
        if need[0] or (has_x2 and need[1]):
            dx1, dx2 = conv_dgrad(dz, w, _spatial(x1), c1, c2, cfg, need[0], has_x2 and need[1], x_dtype=x1.dtype)
        dw = db = None
        pre_db = db_in if cfg.mode == "in" else None
        if pre_db is not None:
            has_b = False                    # (the weight-gradient call below then skips its bias work)
        if need[2] or (has_b and need[3]):
            c1_ = x1.shape[1] + (x2.shape[1] if has_x2 else 0)
            generic = c1_ > 2 and not _thin_out(w, cfg, has_x2)       # the paths that fill a caller-given tensor
            tgt = _direct_grad_target(w) if need[2] else None
            if tgt is not None:
                param, owner, gview = tgt

                def run_wgrad():
                    if generic:
                        _, db_ = conv_wgrad(dz, x1, x2, w, cfg, has_b, out=gview)
                    else:   # thin-input / thin-output kernels return their own tensor: copied into the slot
                        dw_, db_ = conv_wgrad(dz, x1, x2, w, cfg, has_b)
                        gview.copy_(dw_)
                    owner.grad_written(param)
                    return db_
                side = owner.wgrad_stream() if (dz.is_cuda and hasattr(owner, "wgrad_stream")) else None
                if side is not None and has_b:
                    # the bias gradient comes back as a tensor: safe on the side stream only if autograd hands it over
                    # (parameter's .grad is None) instead of accumulating into it on the main stream
                    bp = ctx.bias_ref() if ctx.bias_ref is not None else None
                    if bp is None or bp.grad is not None:
                        side = None
                if side is not None:
                    # the weight gradient does not feed the layer below: launch it on the owner's side stream, behind an
                    # event that marks dz complete; the main stream goes on with the input-gradient chain
                    ready = torch.cuda.Event()
                    ready.record()
                    owner.note_side_launch(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        side.wait_event(ready)
                        db = run_wgrad()
                    for t in (dz, x1, x2):
                        if t is not None:
                            t.record_stream(side)     # their memory is not recycled before the side stream is done
                else:
                    db = run_wgrad()
            else:
                if _direct_grad and need[2]:
                    ent = _direct_entry(w)
                    if ent is not None and hasattr(ent[1], "join_side"):
                        ent[1].join_side()   # autograd will accumulate into a slot a side-stream kernel may be writing
                dw, db = conv_wgrad(dz, x1, x2, w, cfg, has_b)
        if pre_db is not None:
            db = pre_db
        return dx1, dx2, dw, db, dp1, dp2, dres, None


def fused_conv3d(x, w, b=None, stride=1, padding=0, *, x2=None, transposed=False, act=ACT_NONE, slope=0.0,
                 se=None, res=None, inorm=None, eps=1e-5, y_fp32=False):
    """Conv3d / ConvTranspose3d with the fused tails of the hot path.

    x2     second tensor of a virtual channel concat (input = cat([x, x2], 1))
    se     (attn_weight (C,C,1,1,1), attn_bias (C,)): y = act(conv * gate + res)
    inorm  (gamma, beta): y = act(InstanceNorm(conv))
    """
    if se is not None and inorm is not None:
        raise ValueError("se and inorm are exclusive")
    mode = "se" if se is not None else ("in" if inorm is not None else "plain")
    if res is not None and mode != "se":
        raise ValueError("res is only fused behind SEGating")
    cfg = ConvCfg(_triple(stride), _triple(padding), bool(transposed), int(act), float(slope), mode, float(eps),
                  bool(y_fp32))
    p1, p2 = (se if se is not None else (inorm if inorm is not None else (None, None)))
    # operand dtype: the thin layers are fp32 kernels; the matrix-core layers follow mixed_precision() / their input
    cin = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
    thin = cin <= 2 or _thin_out(w, cfg, x2 is not None)
    if thin and x.dtype == torch.bfloat16 and mode == "plain" and act == ACT_NONE and \
            get_backend().thin5_supported(tuple(x.shape), tuple(w.shape), cfg.pad):
        thin = False   # sr_head.2 with bf16 features: the matrix-core kernels of thin_conv_bf16.hip (fp32 result)
    if thin:   # fp32 kernels: bf16 features are cast up, anything else (the fp64 host-logic tests) keeps its dtype
        want = torch.float32 if x.dtype == torch.bfloat16 else x.dtype
    else:
        want = torch.bfloat16 if (_mixed or x.dtype == torch.bfloat16) else x.dtype
    if want in (torch.float32, torch.bfloat16):
        x = x if x.dtype == want else x.to(want)
        x2 = x2 if (x2 is None or x2.dtype == want) else x2.to(want)
        res = res if (res is None or res.dtype == want) else res.to(want)
    # weights that take no gradient in this pass (frozen, or nothing is recorded): the backend may keep their kernel
    # forms between launches (hip_backend, "weight forms"; the backward pass of a recorded node says the same for itself)
    get_backend().keep_forms = not (w.requires_grad and torch.is_grad_enabled())
    return _FusedConv.apply(x, x2, w, b, p1, p2, res, cfg)


class _UpsampleDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Do):
        x = to_cl(x)
        ctx.Di = x.shape[2]
        return get_backend().upsample_depth_fwd(x, Do)

    @staticmethod
    def backward(ctx, dy):
        return get_backend().upsample_depth_bwd(to_cl(dy), ctx.Di), None


class _UpMixDepth(torch.autograd.Function):
    """y = act(bias + sum over depth taps of the depth-interpolated per-tap responses g) -- see rehr_upmix_depth_fwd_f32."""

    @staticmethod
    def forward(ctx, g, bias, Do, Cc, KD, pd, act, slope):
        g = to_cl(g)
        y = get_backend().upmix_depth_fwd(g, bias, Do, Cc, KD, pd, act, slope)
        ctx.save_for_backward(y)
        ctx.cfg = (g.shape[2], KD, pd, act, slope, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        Di, KD, pd, act, slope, has_bias = ctx.cfg
        be = get_backend()
        dy = to_cl(dy)
        db = be.channel_sum_actgrad(dy, y, act, slope) if (has_bias and ctx.needs_input_grad[1]) else None
        return be.upmix_depth_bwd(dy, y, Di, KD, pd, act, slope), db, None, None, None, None, None, None


def upsample_conv3d_depth(x, w, b, scale, act=ACT_NONE, slope=0.0):
    """conv3d(upsample_depth(x, scale), w, b, stride 1, 'same' padding) + activation without the upsampled tensor
    (models/seg_model.py:204-205).  Interpolation along depth and the convolution are both linear: the (kH,kW) part
    of every depth tap runs on the low-resolution slices -- one (1,kH,kW) conv with KD*Cout output channels, scale x
    fewer multiplications -- and one HBM-bound pass interpolates and sums the depth taps.  Same result up to fp32
    reassociation."""
    Cout, Cin, KD, KH, KW = w.shape
    if KD % 2 == 0 or KH % 2 == 0 or KW % 2 == 0 or Cout % 4:
        raise NotImplementedError("upsample_conv3d_depth: odd kernel extents and Cout % 4 == 0")
    Do = int(x.shape[2] * scale)
    wg = w.permute(2, 0, 1, 3, 4).reshape(KD * Cout, Cin, 1, KH, KW)  # row kd*Cout + co
    g = fused_conv3d(x, wg, None, 1, (0, KH // 2, KW // 2))
    # mixed precision: the per-tap responses come back bf16 and the upsampled features stay bf16 (rehr_upmix_depth_*_bf16)
    return _UpMixDepth.apply(g, b, Do, Cout, KD, KD // 2, int(act), float(slope))


def upsample_depth(x, scale):
    """Linear interpolation along depth only, align_corners=True (seg_model.py:204)."""
    Do = int(x.shape[2] * scale)
    return _UpsampleDepth.apply(x, Do)


class _UasrMix(torch.autograd.Function):
    """Softmax blend of the K candidate pairs + uncertainty of the UASR head -- see rehr_uasr_mix_fwd_f32."""

    @staticmethod
    def forward(ctx, om, ue, wu, bu, D):
        om, ue = to_cl(om), to_cl(ue)
        wu, bu = wu.reshape(-1).contiguous(), bu.reshape(-1).contiguous()
        out, unc = get_backend().uasr_mix_fwd(om, ue, wu, bu, D)
        ctx.save_for_backward(om, ue, wu, bu)
        ctx.D = D
        return out, unc

    @staticmethod
    def backward(ctx, gout, gunc):
        om, ue, wu, bu = ctx.saved_tensors
        gout, gunc = gout.contiguous(), gunc.contiguous()  # (an unused output arrives as zeros)
        dom, due, dwu, dbu = get_backend().uasr_mix_bwd(om, ue, wu, bu, gout.to(om.dtype), gunc.to(om.dtype), ctx.D)
        return dom, due, dwu, dbu, None


def uasr_mix_supported(om, ue, n_outputs):
    """The fused UASR head covers K = channels of `ue` per output slice in {4, 8, 16, 32}, two channels per candidate."""
    return (om.dim() == 5 and om.shape[2] == 1 and ue.shape[1] % n_outputs == 0 and om.shape[1] == 2 * ue.shape[1]
            and ue.shape[1] // n_outputs in (4, 8, 16, 32))


def uasr_mix(om, ue, wu, bu, n_outputs):
    """(out (N,2,n_outputs,H,W), unc (N,1,n_outputs,H,W)) of FLAVR's UASR head (reference FLAVR_arch.py:203-246) from the
    two 1x1 responses on the fused slice, om (N, n_outputs*2K, 1, H, W) and ue (N, n_outputs*K, 1, H, W), and
    uncertainty_out's weight (1,K,1,1,1) / bias.  Computed in fp32 (bf16 responses are widened first)."""
    wide = lambda t: t.float() if t.dtype in (torch.bfloat16, torch.float16) else t  # noqa: E731
    return _UasrMix.apply(wide(om), wide(ue), wide(wu).reshape(-1), wide(bu).reshape(-1), int(n_outputs))
