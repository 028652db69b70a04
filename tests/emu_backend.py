"""CPU emulation of the C-ABI entry points (include/rehrseg_hip.h), TEST ONLY.

It restates the *contract* of every entry point -- the descriptor semantics the
HIP kernels implement -- in plain torch on the CPU, so that the host logic in
rehrseg_amd/ops.py (stride phases, tap tables, packing, virtual concat, SE /
InstanceNorm backward algebra) can be checked against torch.nn.functional without
a GPU.  Nothing under rehrseg_amd/ imports this file.
"""
import torch

name = "emu"


def _act(v, act, slope):
    if act == 1:
        return torch.relu(v)
    if act == 2:
        return torch.where(v > 0, v, v * slope)
    return v


def _act_grad(y, act, slope):
    if act == 1:
        return (y > 0).to(y.dtype)
    if act == 2:
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
    return torch.ones_like(y)


def new_act(N, Cc, D, H, W, like, zero=False):
    t = torch.empty((N, Cc, D, H, W), dtype=like.dtype, device=like.device, memory_format=torch.channels_last_3d)
    return t.zero_() if zero else t.fill_(float("nan"))  # unwritten voxels must be noticed


def pack_weights(w, A, Apad, B, T, transpose):
    w = w.contiguous()
    out = torch.zeros((T, Apad, B), dtype=w.dtype)
    if transpose:  # in[b][a][t]
        out[:, :A, :] = w.reshape(B, A, T).permute(2, 1, 0)
    else:          # in[a][b][t]
        out[:, :A, :] = w.reshape(A, B, T).permute(2, 0, 1)
    return out


def _axis(L, s, b, off, size):
    idx = torch.arange(L) * s + b + off
    ok = (idx >= 0) & (idx < size)
    return idx.clamp(0, size - 1), ok


def _gather(X, lattice, s, b, off, dims):
    """X: (N,C,D,H,W) -> (N,C,Ld,Lh,Lw) of X[o*s+b+off], zero outside."""
    (idd, okd), (idh, okh), (idw, okw) = (_axis(lattice[a], s[a], b[a], off[a], dims[a]) for a in range(3))
    g = X[:, :, idd][:, :, :, idh][:, :, :, :, idw]
    m = (okd[:, None, None] & okh[None, :, None] & okw[None, None, :]).to(X.dtype)
    return g * m


def gather_gemm(x1, x2, c1, src_dims, Cin, lattice, s, b, taps, KH, KW, wp, Npad, y, y_dims, Cout,
                os_, ob, bias, act, slope, stats, stats_mode, tile):
    assert Cin % 16 == 0 and (x2 is None or c1 % 32 == 0) and Npad % 32 == 0 and wp.shape[1] == Npad and wp.shape[2] == Cin
    assert tile == (0, 0, 0) or tile[0] * tile[1] * tile[2] == 128
    assert tuple(x1.shape[2:]) == tuple(src_dims) and tuple(y.shape[2:]) == tuple(y_dims)
    X = torch.cat([x1, x2], 1) if x2 is not None else x1
    assert X.shape[1] == Cin
    N = X.shape[0]
    acc = torch.zeros((N, Cout) + tuple(lattice), dtype=X.dtype)
    (cd, od0, ods, kd0, kds), (ch, oh0, ohs, kh0, khs), (cw, ow0, ows, kw0, kws) = taps
    for jd in range(cd):
        for jh in range(ch):
            for jw in range(cw):
                off = (od0 + ods * jd, oh0 + ohs * jh, ow0 + ows * jw)
                wt = ((kd0 + kds * jd) * KH + (kh0 + khs * jh)) * KW + (kw0 + kws * jw)
                g = _gather(X, lattice, s, b, off, src_dims)
                acc += torch.einsum("ncdhw,oc->nodhw", g, wp[wt, :Cout, :])
    if bias is not None:
        acc += bias.view(1, -1, 1, 1, 1)
    acc = _act(acc, act, slope)
    for a in range(3):
        assert ob[a] >= 0 and (lattice[a] - 1) * os_[a] + ob[a] < y_dims[a]
    y[:, :, ob[0]:ob[0] + (lattice[0] - 1) * os_[0] + 1:os_[0],
      ob[1]:ob[1] + (lattice[1] - 1) * os_[1] + 1:os_[1],
      ob[2]:ob[2] + (lattice[2] - 1) * os_[2] + 1:os_[2]] = acc
    if stats_mode:
        stats[:, :, 0] += acc.double().sum((2, 3, 4))
        if stats_mode == 2:
            stats[:, :, 1] += (acc.double() ** 2).sum((2, 3, 4))


def gather_gemm_multi(calls):
    assert 1 <= len(calls) <= 8
    for a in calls:
        assert a[0] is calls[0][0] and a[11] is calls[0][11]  # same x1 / wp (y may differ: split-K slabs)
        gather_gemm(*a)


def sum_slabs_bias_act(slabs, S, bias, act, slope, stats=None):
    SN = slabs.shape[0]
    v = slabs.reshape(S, SN // S, *slabs.shape[1:]).sum(0)
    if bias is not None:
        v = v + bias.view(1, -1, 1, 1, 1)
    v = _act(v, act, slope).contiguous(memory_format=torch.channels_last_3d)
    if stats is not None:
        stats[..., 0] += v.double().sum((2, 3, 4))
        stats[..., 1] += (v.double() ** 2).sum((2, 3, 4))
    return v


def wgrad(l, Ca, g, Cg, N, lattice, g_dims, s, b, taps, KH, KW, dst, dst_off, dst_strides, accumulate, dbias):
    assert Ca % 4 == 0 and Cg % 4 == 0 and tuple(l.shape[2:]) == tuple(lattice)
    flat = dst.view(-1)
    (cd, od0, ods, kd0, kds), (ch, oh0, ohs, kh0, khs), (cw, ow0, ows, kw0, kws) = taps
    sa, sc, st = dst_strides
    ia = torch.arange(Ca)[:, None] * sa
    ic = torch.arange(Cg)[None, :] * sc
    for jd in range(cd):
        for jh in range(ch):
            for jw in range(cw):
                off = (od0 + ods * jd, oh0 + ohs * jh, ow0 + ows * jw)
                wt = ((kd0 + kds * jd) * KH + (kh0 + khs * jh)) * KW + (kw0 + kws * jw)
                gg = _gather(g, lattice, s, b, off, g_dims)
                val = torch.einsum("nadhw,ncdhw->ac", l, gg)
                idx = (dst_off + ia + ic + wt * st).reshape(-1)
                if accumulate:
                    flat[idx] += val.reshape(-1)
                else:
                    flat[idx] = val.reshape(-1)
    if dbias is not None:
        v = l.sum((0, 2, 3, 4))
        dbias.copy_(dbias + v if accumulate else v)


def small_cin_fwd(x, w, bias, y, stride, pad, act, slope, stats, stats_mode):
    assert x.shape[1] <= 2 and w.shape[0] in (16, 32, 64)
    v = torch.nn.functional.conv3d(x, w, bias, stride, pad)
    v = _act(v, act, slope)
    y.copy_(v)
    if stats_mode:
        stats[:, :, 0] += v.double().sum((2, 3, 4))
        if stats_mode == 2:
            stats[:, :, 1] += (v.double() ** 2).sum((2, 3, 4))


def im2col(x, w, out_dims, stride, pad, Kpad):
    N, Cin = x.shape[:2]
    KD, KH, KW = w.shape[2:]
    T = KD * KH * KW
    assert Kpad % 4 == 0 and Kpad >= Cin * T
    col = torch.zeros((N, Kpad) + tuple(out_dims), dtype=x.dtype)
    for ci in range(Cin):
        for kd in range(KD):
            for kh in range(KH):
                for kw in range(KW):
                    k = ci * T + (kd * KH + kh) * KW + kw
                    col[:, k:k + 1] = _gather(x[:, ci:ci + 1], out_dims, stride, tuple(-p for p in pad),
                                              (kd, kh, kw), tuple(x.shape[2:]))
    return col.contiguous(memory_format=torch.channels_last_3d)


def small_cin_wgrad(x, w, dy, stride, pad, want_bias):
    dw = torch.nn.grad.conv3d_weight(x, w.shape, dy, stride, pad)
    return dw, (dy.sum((0, 2, 3, 4)) if want_bias else None)


def small_cout_fwd(x, w, bias, y, pad, act, slope):
    assert w.shape[0] <= 4 and x.shape[1] % 16 == 0
    y.copy_(_act(torch.nn.functional.conv3d(x, w, bias, 1, pad), act, slope))


def small_cout_dgrad(dy, w, x_shape, pad):
    return torch.nn.grad.conv3d_input(x_shape, w, dy, 1, pad).contiguous(memory_format=torch.channels_last_3d)


def small_cout_wgrad(x, w, dy, pad, want_bias):
    return torch.nn.grad.conv3d_weight(x, w.shape, dy, 1, pad), (dy.sum((0, 2, 3, 4)) if want_bias else None)


def se_gate_fwd(stats, w, b, N, Cc, S):
    mean = (stats[:, :, 0] / S).to(w.dtype)
    gate = torch.sigmoid(mean @ w.t() + b)
    return gate, mean


def scale_res_act_fwd(x, gate, res, act, slope):
    v = x * gate[:, :, None, None, None]
    if res is not None:
        v = v + res
    return _act(v, act, slope).contiguous(memory_format=torch.channels_last_3d)


def scale_res_act_bwd(dy, y, x, gate, want_dres, act, slope):
    dz = dy * _act_grad(y, act, slope)
    dx = (dz * gate[:, :, None, None, None]).contiguous(memory_format=torch.channels_last_3d)
    dgate = (dz * x).double().sum((2, 3, 4))
    return dx, (dz.contiguous(memory_format=torch.channels_last_3d) if want_dres else None), dgate


def se_gate_bwd(dgate, gate, mean, w, S):
    ds = dgate.to(gate.dtype) * gate * (1 - gate)
    dw = ds.t() @ mean
    db = ds.sum(0)
    k = (ds @ w) / S
    return dw, db, k


def add_channel_const(x, k):
    x += k[:, :, None, None, None]


def instnorm_act_fwd(x, stats, gamma, beta, eps, act, slope):
    N, Cc = x.shape[:2]
    S = x[0, 0].numel()
    mean = stats[:, :, 0] / S
    var = (stats[:, :, 1] / S - mean * mean).clamp_min(0)
    rstd = 1.0 / torch.sqrt(var + eps)
    mr = torch.stack([mean, rstd], -1).to(x.dtype)
    xh = (x - mr[:, :, 0, None, None, None]) * mr[:, :, 1, None, None, None]
    y = _act(xh * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1), act, slope)
    return y.contiguous(memory_format=torch.channels_last_3d), mr


def instnorm_act_bwd(dy, x, mr, gamma, beta, act, slope):
    xh = (x - mr[:, :, 0, None, None, None]) * mr[:, :, 1, None, None, None]
    z = xh * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)
    dz = dy * _act_grad(z, act, slope)
    dgamma = (dz * xh).sum((0, 2, 3, 4))
    dbeta = dz.sum((0, 2, 3, 4))
    m1 = dz.mean((2, 3, 4), keepdim=True)
    m2 = (dz * xh).mean((2, 3, 4), keepdim=True)
    dx = mr[:, :, 1, None, None, None] * gamma.view(1, -1, 1, 1, 1) * (dz - m1 - xh * m2)
    return dx.contiguous(memory_format=torch.channels_last_3d), dgamma, dbeta


def _depth_w(Di, Do, dtype):
    Wm = torch.zeros((Do, Di), dtype=dtype)
    scale = (Di - 1) / (Do - 1) if Do > 1 else 0.0
    for od in range(Do):
        src = torch.tensor(scale, dtype=torch.float32) * od
        i0 = min(int(src), Di - 1)
        i1 = min(i0 + 1, Di - 1)
        w1 = float(src - i0)
        Wm[od, i0] += 1 - w1
        Wm[od, i1] += w1
    return Wm


def upsample_depth_fwd(x, Do):
    Wm = _depth_w(x.shape[2], Do, x.dtype)
    return torch.einsum("od,ncdhw->ncohw", Wm, x).contiguous(memory_format=torch.channels_last_3d)


def upsample_depth_bwd(dy, Di):
    Wm = _depth_w(Di, dy.shape[2], dy.dtype)
    return torch.einsum("od,ncohw->ncdhw", Wm, dy).contiguous(memory_format=torch.channels_last_3d)


def _upmix_w(Di, Do, KD, pd, dtype):
    """M[kd][od][j]: coefficient of low-resolution slice j in output slice od through depth tap kd."""
    Wm = _depth_w(Di, Do, dtype)  # [ud][j]
    M = torch.zeros((KD, Do, Di), dtype=dtype)
    for kd in range(KD):
        for od in range(Do):
            ud = od + kd - pd
            if 0 <= ud < Do:
                M[kd, od] = Wm[ud]
    return M


def upmix_depth_fwd(g, bias, Do, Cc, KD, pd, act, slope):
    N, _, Di, H, W = g.shape
    M = _upmix_w(Di, Do, KD, pd, g.dtype)
    y = torch.einsum("koj,nkcjhw->ncohw", M, g.reshape(N, KD, Cc, Di, H, W))
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1, 1)
    return _act(y, act, slope).contiguous(memory_format=torch.channels_last_3d)


def channel_sum_actgrad(dy, y, act, slope):
    return (dy * _act_grad(y, act, slope)).sum(dim=(0, 2, 3, 4))


def upmix_depth_bwd(dy, y, Di, KD, pd, act, slope):
    dz = dy * _act_grad(y, act, slope)
    N, Cc, Do, H, W = dz.shape
    M = _upmix_w(Di, Do, KD, pd, dz.dtype)
    dg = torch.einsum("koj,ncohw->nkcjhw", M, dz).reshape(N, KD * Cc, Di, H, W)
    return dg.contiguous(memory_format=torch.channels_last_3d)


def _uasr_parts(om, ue, wu, bu, D):
    N, _, _, H, W = om.shape
    K = ue.shape[1] // D
    o = om[:, :, 0].reshape(N, D, K, 2, H, W)
    s = torch.softmax(ue[:, :, 0].reshape(N, D, K, H, W), dim=2)
    return o[:, :, :, 0], o[:, :, :, 1], s, wu.reshape(1, 1, K, 1, 1), K


def uasr_mix_fwd(om, ue, wu, bu, D):
    a, b, s, w, _ = _uasr_parts(om, ue, wu, bu, D)
    out = torch.stack([(s * (torch.tanh(a) + 1) / 2).sum(2), (s * b).sum(2)], dim=1)
    return out, torch.sigmoid((s * w).sum(2) + bu.reshape(1, 1, 1, 1)).unsqueeze(1)


def uasr_mix_bwd(om, ue, wu, bu, gout, gunc, D):
    a, b, s, w, K = _uasr_parts(om, ue, wu, bu, D)
    N, _, _, H, W = om.shape
    g0, g1 = gout[:, 0].unsqueeze(2), gout[:, 1].unsqueeze(2)   # (N, D, 1, H, W)
    th = torch.tanh(a)
    img = (th + 1) / 2
    z = (s * w).sum(2, keepdim=True)
    u = torch.sigmoid(z + bu.reshape(1, 1, 1, 1, 1))
    gz = gunc[:, 0].unsqueeze(2) * u * (1 - u)
    ds = g0 * img + g1 * b + gz * w
    due = s * (ds - (s * ds).sum(2, keepdim=True))
    dom = torch.stack([g0 * s * (1 - th * th) / 2, g1 * s], dim=3)  # (N, D, K, 2, H, W)
    cl = torch.channels_last_3d
    return (dom.reshape(N, D * K * 2, 1, H, W).contiguous(memory_format=cl),
            due.reshape(N, D * K, 1, H, W).contiguous(memory_format=cl),
            (gz * s).sum(dim=(0, 1, 3, 4)), gz.sum().reshape(1))


def window_stem_assemble(g, mean, bias, B, nwin, act, slope):
    S, Cc, _, h, w = g[0].shape
    ns = nwin + 3
    gs = [t[:-1, :, 0].reshape(B, ns, Cc, h, w) for t in g]
    rs = [t[-1, :, 0] for t in g]
    out = torch.zeros(B, nwin, Cc, 4, h, w, dtype=g[0].dtype)
    m = mean.reshape(B, nwin, 1, 1, 1)
    for k in range(4):
        for kd in range(3):
            ks = k + kd - 1
            if 0 <= ks <= 3:
                out[:, :, :, k] += gs[kd][:, ks:ks + nwin] - m * rs[kd]
    if bias is not None:
        out = out + bias.view(1, 1, -1, 1, 1, 1)
    return _act(out.reshape(B * nwin, Cc, 4, h, w), act, slope).contiguous(memory_format=torch.channels_last_3d)


def act_fwd(x, act, slope):
    return _act(x, act, slope)


def act_bwd(dy, y, act, slope):
    return dy * _act_grad(y, act, slope)


def channel_sum(x):
    return x.sum((0, 2, 3, 4))
