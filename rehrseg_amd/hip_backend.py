"""Launch layer: turns torch tensors into the plain-pointer descriptors of the
C-ABI (include/rehrseg_hip.h) and launches on torch's current HIP stream.

Every activation handed to this module is a 5-D torch tensor with logical shape
(N, C, D, H, W) whose memory is NDHWC-dense (``ops.to_cl``).  PyTorch is used for
device memory and streams only; all arithmetic happens in librehrseg_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
import functools
import weakref

import torch
from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook

from . import lib as L

name = "hip"


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-launch timing of the MFMA kernels (bench.py's roofline line): HIP
# events recorded on the launch stream around every gather-GEMM / wgrad launch.
_prof = None


def profile_start():
    global _prof
    _prof = []


def profile_stop(layers=False):
    """-> {family: {"flops", "seconds", "launches"}} summed over the recorded launches; layers=True: the launches one by
    one as (family, shape tag, flops, seconds) instead (tools/layer_times.py)."""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    if layers:
        return [(fam, tag, flops, e0.elapsed_time(e1) * 1e-3) for fam, flops, e0, e1, tag in rec or []]
    out = {}
    for fam, flops, e0, e1, _ in rec or []:
        d = out.setdefault(fam, {"flops": 0.0, "seconds": 0.0, "launches": 0})
        d["flops"] += flops
        d["seconds"] += e0.elapsed_time(e1) * 1e-3
        d["launches"] += 1
    return out


@functools.lru_cache(maxsize=4096)
def _reach(L, s, b, taps, size):
    """Number of (lattice point, tap) pairs of one axis whose source index is in bounds:
    the algorithmic work of a launch excludes taps that fall into the zero padding."""
    cnt, off0, offs, _, _ = taps
    tot = 0
    for j in range(cnt):
        c = b + off0 + offs * j
        lo = max(0, -(c // s)) if c < 0 else 0          # smallest o with o*s + c >= 0
        hi = min(L - 1, (size - 1 - c) // s) if size - 1 - c >= 0 else -1
        tot += max(0, hi - lo + 1)
    return tot


def _algo_flops(N, lattice, s, b, taps, dims, c_a, c_b):
    f = 2.0 * N * c_a * c_b
    for a in range(3):
        f *= _reach(lattice[a], s[a], b[a], tuple(taps[a]), dims[a])
    return f


class _timed:
    def __init__(self, fam, flops, tag=None):
        self.fam, self.flops, self.tag = fam, flops, tag

    def __enter__(self):
        if _prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _prof is not None:
            self.e1.record()
            _prof.append((self.fam, self.flops, self.e0, self.e1, self.tag() if callable(self.tag) else self.tag))
        return False


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _taps(t):
    return L.AxisTaps(*t)


def _chk_dev(*ts, f64=()):
    """Every operand of the fp32 kernels must be a float32 device tensor; `f64` lists the statistics buffers
    (double accumulators) of the call."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise L.RehrsegHipError("librehrseg_hip.so needs device tensors (no CPU fallback)")
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise L.RehrsegHipError(f"unsupported dtype {t.dtype} (the kernels read float32, the mixed-precision "
                                    "variants bfloat16)")
    for t in f64:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float64:
            raise L.RehrsegHipError("statistics buffers are float64 device tensors")


def _fn(base, t):
    """C entry point `base`_f32 or `base`_bf16 by the dtype of activation tensor `t`."""
    name = base + ("_bf16" if t.dtype == torch.bfloat16 else "_f32")
    return getattr(L.load(), name), name


def _same_dtype(*ts):
    dts = {t.dtype for t in ts if t is not None}
    if len(dts) > 1:
        raise L.RehrsegHipError(f"activation operands of one call must share a dtype, got {sorted(map(str, dts))}")


# Small zero-initialised fp64 accumulators (per-(sample, channel) statistics, gradient reductions): every fused layer
# needs one or two per pass, and a torch.zeros() each is a 5 us fill launch -- ~90 per cfg-2 step.  They are carved out
# of a pooled chunk that is zeroed by ONE fill when it is allocated; a slice is handed out once and never reused (the
# chunk lives as long as any slice of it does), so no kernel ever sees stale values.  One pool per (device, stream): the
# fill runs on the stream that is current when the chunk is made, and only kernels of that stream may rely on it.
_zero_pool = {}
_ZERO_CHUNK = 1 << 16   # doubles (512 KB)


def zeros_f64(shape, device):
    n = 1
    for s in shape:
        n *= int(s)
    n_al = (n + 1) & ~1                                   # keep 16-byte alignment of every slice
    if n_al > _ZERO_CHUNK // 4:
        return torch.zeros(shape, dtype=torch.float64, device=device)
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ent = _zero_pool.get(key)
    if ent is None or ent[1] + n_al > _ZERO_CHUNK:
        ent = [torch.zeros(_ZERO_CHUNK, dtype=torch.float64, device=device), 0]
        _zero_pool[key] = ent
    out = ent[0][ent[1]:ent[1] + n].view(shape)
    ent[1] += n_al
    return out


def new_act(N, Cc, D, H, W, like, zero=False, dtype=None):
    """NDHWC activation in `like`'s dtype (fp32, or bf16 on the mixed-precision path) unless `dtype` says otherwise."""
    dt = dtype if dtype is not None else (like.dtype if like.dtype == torch.bfloat16 else torch.float32)
    t = torch.empty((N, Cc, D, H, W), dtype=dt, device=like.device, memory_format=torch.channels_last_3d)
    return t.zero_() if zero else t


# ---- weight forms kept across launches ----------------------------------------------------------------------------
# A convolution needs its weights as a kernel panel (pack_weights) and, on the fp32 Winograd kernels, as
# transform-domain fragments (the library's weight-transform launch into `wino_ws`): two 5-20 us launches in front of
# every forward and every input-gradient kernel.  Both depend on the parameter's VALUES and the layer geometry only.
# For weights that take no gradient in the pass at hand -- frozen parameters, passes under torch.no_grad(): the joint
# step's teacher, the tiled predictor walking hundreds of tiles, validation -- they are kept per (parameter, geometry)
# and rebuilt, where they are needed, when the
# parameter has changed: its version counter has moved (load_state_dict, any in-place torch op) or an optimizer has
# stepped over it (torch's fused optimizers do NOT move version counters, so a global optimizer-step post hook stamps
# the parameters of every optimizer that steps).  A write through `.data` is seen by neither, by torch's design: call
# invalidate_weight_forms() after one.
# Trainable parameters in a recorded pass are NOT kept: every form would be stale once per step anyway.  Rebuilding all
# stale forms of a step in one batch in front of the first convolution (on one side stream next to the forward, or
# fanned out over 2 / 4 / 8 streams and joined) was measured and is SLOWER than the launches where they stand
# (profiles/r03_ab_weight_forms.txt: cfg-2 +0.4..0.6 ms, cfg-3 +0.6..1.0 ms): in line, each small launch runs in the
# shadow of the matrix-core kernels around it, whose clocks the power budget sets -- the 1.5 ms these launches sum to in
# a kernel trace is not on the step's critical path.
WEIGHT_FORM_CACHE = True     # False: every call packs / transforms for itself (tests compare the two)


def _after_optimizer_step(opt, args, kwargs):
    for g in opt.param_groups:
        for p in g["params"]:
            p._rehr_stamp = getattr(p, "_rehr_stamp", 0) + 1


_register_step_hook(_after_optimizer_step)


def _ver(p):
    """What a form remembers of the parameter it was made from."""
    return (p._version, getattr(p, "_rehr_stamp", 0))


keep_forms = False           # set by ops.fused_conv3d / _FusedConv.backward in front of every launch sequence: the weight of this node takes
                             # no gradient in this pass (frozen parameter, or nothing is being recorded)


def _keeps_forms(w):
    return WEIGHT_FORM_CACHE and keep_forms and isinstance(w, torch.nn.Parameter)


class _Form:
    __slots__ = ("param", "version", "out", "make", "stream", "readers", "dep", "__weakref__")

    def __init__(self, key, param, make, dep=None):
        self.param = weakref.ref(param, lambda _r, k=key: _drop_form(k))   # the forms go when the parameter goes
        self.version = None          # _ver(parameter) `out` was made from
        self.out = None              # pack: the panel tensor; Winograd: list of scratch tensors (one per descriptor)
        self.make = make             # make(parameter): launches the rebuild on the current stream
        self.stream = None           # the stream of the last rebuild
        self.readers = set()         # streams that have consumed `out` since (a rebuild has to wait for them)
        self.dep = dep               # Winograd forms: the pack form they read


_forms = {}                # key -> _Form
_form_of_panel = {}        # id(panel tensor) -> pack form (Winograd forms find the parameter behind a `wp`)
form_rebuilds = 0          # launches issued for (re)building kept forms (tests look at this)


def _drop_form(key):
    f = _forms.pop(key, None)
    if f is not None and key[0] == "pack":
        _form_of_panel.pop(id(f.out), None)


def invalidate_weight_forms():
    """Forget every kept weight form (after writing parameters through `.data`, which no version counter sees)."""
    _forms.clear()
    _form_of_panel.clear()


def _form_ready(f, p):
    """Make form `f` of parameter `p` current and order the current stream behind its last rebuild."""
    global form_rebuilds
    cur = torch.cuda.current_stream(p.device)
    v = _ver(p)
    if f.version != v:
        if f.dep is not None:
            _form_ready(f.dep, p)
        for r in f.readers:                             # nobody may still be reading what is overwritten
            if r != cur:
                cur.wait_stream(r)
        if f.stream is not None and f.stream != cur:
            cur.wait_stream(f.stream)
        f.make(p)
        form_rebuilds += 1
        f.version, f.stream, f.readers = v, cur, set()
    if f.stream != cur and cur not in f.readers:        # made on another stream: once per rebuild and consumer stream
        cur.wait_stream(f.stream)
    f.readers.add(cur)


def _pack_launch(src, out, A, Apad, B, T, transpose):
    if out.dtype == torch.bfloat16:
        L.check(L.load().rehr_pack_weights_bf16(_ptr(src), _ptr(out), A, Apad, B, T, int(transpose), _stream()),
                "rehr_pack_weights_bf16")
    else:
        L.check(L.load().rehr_pack_weights_f32(_ptr(src), _ptr(out), A, Apad, B, T, int(transpose), _stream()),
                "rehr_pack_weights_f32")


def pack_weights(w, A, Apad, B, T, transpose, dtype=torch.float32, part=None):
    """fp32 master weights in the torch parameter layout -> kernel panel [T][Apad][B] in `dtype` (fp32 / bf16).
    `part` = (dim, lo, count): the panel of w.narrow(dim, lo, count) (the halves of a virtual channel concat).
    Panels of frozen parameters / of no-grad passes are kept until the parameter changes ("weight forms" above)."""
    _chk_dev(w)
    if w.dtype != torch.float32:
        raise L.RehrsegHipError("master weights are float32")

    def src(p):
        return (p if part is None else p.narrow(*part)).contiguous()

    if not _keeps_forms(w):
        out = torch.empty((T, Apad, B), dtype=dtype, device=w.device)
        _pack_launch(src(w), out, A, Apad, B, T, transpose)
        return out
    key = ("pack", id(w), tuple(w.shape), part, A, Apad, B, T, bool(transpose), dtype)
    f = _forms.get(key)
    if f is None or f.param() is not w:
        out = torch.empty((T, Apad, B), dtype=dtype, device=w.device)
        f = _forms[key] = _Form(key, w, lambda p: _pack_launch(src(p), out, A, Apad, B, T, transpose))
        f.out = out
        _form_of_panel[id(out)] = f
    _form_ready(f, w)
    return f.out


# Kernel selection travels in the descriptors (the library reads no environment).  With `flags` / `debug_flags` = 0
# the library picks its measured-best kernels: forward / input gradient take the Winograd kernels when the descriptor
# carries scratch.  The switches below set `debug_flags` bits (unstable, include/rehrseg_hip.h): tests flip them to
# compare kernel organisations with each other (transform-domain against direct, brick against per-tap, ...).
USE_WINOGRAD = True
USE_WINOGRAD_WGRAD = True   # False -> REHR_DBG_WGRAD_DIRECT
USE_HALO_BF16 = True   # mixed precision: LDS halo-brick kernel for unit-stride 3x3(x3) taps (False -> REHR_DBG_GG_NO_HALO)
WGRAD_TAP_COLOCATE = True   # Winograd weight gradient: the depth taps of a split on one XCD (shared dY / x in L2)
USE_WGRAD_TAP_SKIP = True   # Winograd weight gradient: a depth tap walks only the slices whose source slice exists
W32P_BLOCKS = 0   # 0: the library picks; 1 / 2 force two 256-thread blocks per CU / one 512-thread block (tests, A/B)
PHASE_INTERLEAVE = False   # multi-phase launches: the phases of a lattice tile as consecutive blocks of one XCD (measured slower)
WINO_BAND_MAJOR = True   # fp32 Winograd tile order: an XCD walks depth inside a band of rows (L2 reuse of the depth taps)
USE_TCONV_KS = True   # kernel == stride transposed convolutions: all phases of an input tile in one block (tconv_ks.hip)
USE_WINO_FLAT8 = True   # fp32 Winograd on planes that 16 x 16 regions tile badly: wino_flat8_conv_kernel
WINO_FLAT8_TILES = 0    # 0: the library picks 32 or 64 tiles per block; 1 / 2 force 32 / 64 (tests, A/B)
wino_wgrad_launches = 0  # weight gradients taken by the Winograd kernel
wino_launches = 0  # contractions handed to the Winograd kernels so far (tests look at this)


def _gg_desc(d, x1, x2, c1, src_dims, Cin, lattice, s, b, taps, KH, KW, wp, Npad, y, y_dims, Cout,
             os_, ob, bias, act, slope, stats, stats_mode, tile):
    _chk_dev(x1, x2, wp, y, bias, f64=(stats,))
    d.x1, d.x2, d.c1 = _ptr(x1), _ptr(x2), c1
    d.ldx1 = x1.shape[1]
    d.ldx2 = x2.shape[1] if x2 is not None else 0
    d.N = x1.shape[0]
    d.Di, d.Hi, d.Wi = src_dims
    d.Cin = Cin
    d.Ld, d.Lh, d.Lw = lattice
    d.sd, d.sh, d.sw = s
    d.bd, d.bh, d.bw = b
    d.td, d.th, d.tw = _taps(taps[0]), _taps(taps[1]), _taps(taps[2])
    d.KH, d.KW = KH, KW
    d.wp, d.Npad = _ptr(wp), Npad
    d.y = _ptr(y)
    d.Dy, d.Hy, d.Wy = y_dims
    d.Cout, d.ldy = Cout, y.shape[1]
    d.osd, d.osh, d.osw = os_
    d.obd, d.obh, d.obw = ob
    d.bias, d.act, d.slope = _ptr(bias), act, slope
    d.stats, d.stats_mode = _ptr(stats), stats_mode
    d.tile_d, d.tile_h, d.tile_w = tile
    d.wino_ws, d.wino_ws_bytes = None, 0
    d.flags = 0
    d.debug_flags = 0
    need = 0
    if x1.dtype == torch.bfloat16:
        if wp.dtype != torch.bfloat16 or (x2 is not None and x2.dtype != torch.bfloat16):
            raise L.RehrsegHipError("mixed-precision gather-GEMM: x1, x2 and the packed weights must all be bfloat16")
        if bias is not None and bias.dtype != torch.float32:
            raise L.RehrsegHipError("bias stays float32")
        d.flags = L.GG_Y_F32 if y.dtype == torch.float32 else 0
        d.debug_flags = (0 if USE_HALO_BF16 else L.DBG_GG_NO_HALO)
    elif wp.dtype != torch.float32 or y.dtype != torch.float32 or (x2 is not None and x2.dtype != torch.float32):
        raise L.RehrsegHipError("fp32 gather-GEMM: every operand must be float32")
    elif USE_WINOGRAD and tile[0] >= 0:
        d.debug_flags = ((0 if USE_WINO_FLAT8 else L.DBG_GG_NO_FLAT8) |
                         {1: L.DBG_GG_W32P_TWO_PER_CU, 2: L.DBG_GG_W32P_ONE_PER_CU}.get(W32P_BLOCKS, 0) |
                         {1: L.DBG_GG_FLAT8_HALF, 2: L.DBG_GG_FLAT8_FULL}.get(WINO_FLAT8_TILES, 0))
        need = L.load().rehr_gather_gemm_wino_bytes(C.byref(d))
    if PHASE_INTERLEAVE:
        d.debug_flags |= L.DBG_GG_INTERLEAVE
    if not USE_TCONV_KS:
        d.debug_flags |= L.DBG_GG_NO_TCONV_KS
    if not WINO_BAND_MAJOR:
        d.debug_flags |= L.DBG_GG_SLICE_MAJOR
    # (the Winograd scratch is attached by _attach_ws once every descriptor of the call is known)
    _gg_desc.need = need
    _gg_desc.sig = (c1, x1.shape[1], x2.shape[1] if x2 is not None else 0, x1.shape[0], tuple(src_dims), Cin,
                    tuple(lattice), tuple(s), tuple(b), tuple(tuple(t) for t in taps), KH, KW, Npad, tuple(y_dims), Cout,
                    y.shape[1], tuple(os_), tuple(ob), bias is not None, act, stats_mode, tuple(tile), d.debug_flags)
    return _algo_flops(d.N, lattice, s, b, taps, src_dims, Cin, Cout) if _prof is not None else 0.0


def _attach_ws(descs, n, needs, sigs, wp):
    """Winograd scratch for the descriptors of one call that want it.  Returns what must stay referenced until the
    launch has been enqueued.  When `wp` is the kept panel of a parameter the transformed weights are kept too (one form
    per call geometry) and the call runs with REHR_GG_WS_READY."""
    if not any(needs):
        return None
    global wino_launches
    wino_launches += sum(1 for nb in needs if nb)
    dev = wp.device
    pf = _form_of_panel.get(id(wp)) if WEIGHT_FORM_CACHE else None
    p = pf.param() if pf is not None else None
    if p is None or not _keeps_forms(p):
        keeps = [torch.empty(nb // 4, dtype=torch.float32, device=dev) if nb else None for nb in needs]
        for i in range(n):
            if needs[i]:
                descs[i].wino_ws, descs[i].wino_ws_bytes = _ptr(keeps[i]), needs[i]
        return keeps
    key = ("ws", id(p), id(pf), tuple(sigs))
    f = _forms.get(key)
    if f is None or f.param() is not p or f.dep is not pf:
        outs = [torch.empty(nb // 4, dtype=torch.float32, device=dev) if nb else None for nb in needs]
        cp = (L.GatherGemmDesc * n)()       # the rebuild call: the same descriptors, transforms only (operand pointers
        for i in range(n):                  # other than wp / wino_ws go stale and are not dereferenced)
            C.memmove(C.byref(cp[i]), C.byref(descs[i]), C.sizeof(L.GatherGemmDesc))
            if needs[i]:
                cp[i].wino_ws, cp[i].wino_ws_bytes = _ptr(outs[i]), needs[i]
            cp[i].flags |= L.GG_WS_ONLY

        def make(_p, cp=cp, n=n):
            L.check(L.load().rehr_gather_gemm_multi_f32(cp, n, _stream()), "rehr_gather_gemm_multi_f32 (weight forms)")

        f = _forms[key] = _Form(key, p, make, dep=pf)
        f.out = outs
    _form_ready(f, p)
    for i in range(n):
        if needs[i]:
            descs[i].wino_ws, descs[i].wino_ws_bytes = _ptr(f.out[i]), needs[i]
        descs[i].flags |= L.GG_WS_READY
    return f.out


def _gg_tag(d):
    return (f"N{d.N} {d.Cin}->{d.Cout} lattice {d.Ld}x{d.Lh}x{d.Lw} stride {d.sd}{d.sh}{d.sw} "
            f"taps {d.td.count}x{d.th.count}x{d.tw.count}" + (" +x2" if d.x2 else ""))


def _gg_family(d):
    """Profiling family of a launch: which kernel family the library picks for this descriptor."""
    if not d.wino_ws:
        return "gather_gemm"
    return "wino_conv" if d.th.count == 3 else "wino22_conv"


def gather_gemm(*args):
    d = L.GatherGemmDesc()
    flops = _gg_desc(d, *args)
    if args[0].dtype == torch.bfloat16:
        with _timed("gather_gemm_bf16", flops, lambda: _gg_tag(d)):
            L.check(L.load().rehr_gather_gemm_bf16(C.byref(d), _stream()), "rehr_gather_gemm_bf16")
        return
    arr = (L.GatherGemmDesc * 1)(d)
    keep = _attach_ws(arr, 1, [_gg_desc.need], [_gg_desc.sig], args[11])
    with _timed(_gg_family(arr[0]), flops, lambda: _gg_tag(arr[0])):
        L.check(L.load().rehr_gather_gemm_multi_f32(arr, 1, _stream()), "rehr_gather_gemm_f32")
    del keep


def gather_gemm_multi(calls):
    """`calls` = argument tuples of gather_gemm for the stride phases of one layer: one grid."""
    if len(calls) == 1:
        return gather_gemm(*calls[0])
    arr = (L.GatherGemmDesc * len(calls))()
    flops, needs, sigs = 0.0, [], []
    for i, a in enumerate(calls):
        flops += _gg_desc(arr[i], *a)
        needs.append(_gg_desc.need)
        sigs.append(_gg_desc.sig)
    if calls[0][0].dtype == torch.bfloat16:
        with _timed("gather_gemm_bf16", flops, lambda: f"{len(calls)} parts of " + _gg_tag(arr[0])):
            L.check(L.load().rehr_gather_gemm_multi_bf16(arr, len(calls), _stream()), "rehr_gather_gemm_multi_bf16")
        return
    keep = _attach_ws(arr, len(calls), needs, sigs, calls[0][11])   # every part its own scratch: ONE launch runs them all
    with _timed(_gg_family(arr[0]) if all(arr[i].wino_ws for i in range(len(calls))) else "gather_gemm", flops,
                lambda: f"{len(calls)} parts of " + _gg_tag(arr[0])):
        L.check(L.load().rehr_gather_gemm_multi_f32(arr, len(calls), _stream()), "rehr_gather_gemm_multi_f32")
    del keep


def sum_slabs_bias_act(slabs, S, bias, act, slope, stats=None, out_dtype=None):
    """slabs: (S*N, C, D, H, W) NDHWC fp32 partial results -> (N, C, D, H, W) = act(bias + sum over S) in `out_dtype`
    (fp32, or bf16 on the mixed-precision path); `stats` (N, C, 2) double, pre-zeroed: also accumulate the per-(n,c)
    sum / sum of squares (of the fp32 sums)."""
    _chk_dev(slabs, bias, f64=(stats,))
    if slabs.dtype != torch.float32:
        raise L.RehrsegHipError("split-K slabs are float32")
    SN, Cc, D, H, W = slabs.shape
    N = SN // S
    bf = out_dtype == torch.bfloat16
    y = new_act(N, Cc, D, H, W, like=slabs, dtype=torch.bfloat16 if bf else torch.float32)
    rows = N * D * H * W
    lib = L.load()
    if stats is not None:
        fn = lib.rehr_sum_slabs_stats_bf16 if bf else lib.rehr_sum_slabs_stats_f32
        L.check(fn(_ptr(slabs), S, rows * Cc, _ptr(bias), _ptr(y), N, D * H * W, Cc, act, slope, _ptr(stats), _stream()),
                "rehr_sum_slabs_stats")
        return y
    fn = lib.rehr_sum_slabs_bias_act_bf16 if bf else lib.rehr_sum_slabs_bias_act_f32
    L.check(fn(_ptr(slabs), S, rows * Cc, _ptr(bias), _ptr(y), rows, Cc, act, slope, _stream()), "rehr_sum_slabs_bias_act")
    return y


def wgrad(l, Ca, g, Cg, N, lattice, g_dims, s, b, taps, KH, KW, dst, dst_off, dst_strides, accumulate, dbias):
    """dst (fp32, the master-weight gradient) from l / g in fp32, or both in bf16 (mixed precision: no dbias)."""
    _chk_dev(l, g, dst, dbias)
    bf16 = l.dtype == torch.bfloat16
    if (g.dtype == torch.bfloat16) != bf16 or dst.dtype != torch.float32:
        raise L.RehrsegHipError("weight gradient: l and g share one dtype (fp32 or bf16), dst is float32")
    d = L.WgradDesc()
    d.l, d.ldl, d.Ca = _ptr(l), l.shape[1], Ca
    d.g, d.ldg, d.Cg = _ptr(g), g.shape[1], Cg
    d.N = N
    d.Ld, d.Lh, d.Lw = lattice
    d.Dg, d.Hg, d.Wg = g_dims
    d.sd, d.sh, d.sw = s
    d.bd, d.bh, d.bw = b
    d.td, d.th, d.tw = _taps(taps[0]), _taps(taps[1]), _taps(taps[2])
    d.KH, d.KW = KH, KW
    d.dst = C.c_void_p(dst.data_ptr() + 4 * dst_off)
    d.dst_sa, d.dst_sc, d.dst_st = dst_strides
    d.accumulate = int(accumulate)
    d.dbias = _ptr(dbias)
    d.flags = 0
    d.debug_flags = ((0 if USE_WINOGRAD_WGRAD else L.DBG_WGRAD_DIRECT) | (0 if USE_WGRAD_TAP_SKIP else L.DBG_WGRAD_NO_TAP_SKIP) |
                     (0 if WGRAD_TAP_COLOCATE else L.DBG_WGRAD_NO_TAP_COLOCATE))
    lib = L.load()
    flops = _algo_flops(N, lattice, s, b, taps, g_dims, Ca, Cg) if _prof is not None else 0.0
    if bf16:
        if dbias is not None:
            raise L.RehrsegHipError("mixed-precision weight gradient has no fused bias gradient (use channel_sum)")
        nbytes = lib.rehr_wgrad_bf16_workspace_bytes(C.byref(d))
        if nbytes < 0:
            L.check(int(nbytes), "rehr_wgrad_bf16_workspace_bytes")
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=l.device)
        d.workspace, d.workspace_bytes = _ptr(ws), nbytes
        with _timed("wgrad_bf16", flops, lambda: f"N{N} {Ca}x{Cg} lattice {tuple(lattice)} stride {tuple(s)} "
                                             f"taps {taps[0][0]}x{taps[1][0]}x{taps[2][0]}"):
            L.check(lib.rehr_wgrad_bf16(C.byref(d), _stream()), "rehr_wgrad_bf16")
        return
    nbytes = lib.rehr_wgrad_workspace_bytes(C.byref(d))
    if nbytes < 0:
        L.check(int(nbytes), "rehr_wgrad_workspace_bytes")
    global wino_wgrad_launches
    wino = int(lib.rehr_wgrad_uses_winograd(C.byref(d)))
    wino_wgrad_launches += wino
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=l.device)
    d.workspace, d.workspace_bytes = _ptr(ws), nbytes
    with _timed(("wino_wgrad" if d.th.count == 3 else "wino22_wgrad") if wino else "wgrad", flops,
                lambda: f"N{N} {Ca}x{Cg} lattice {tuple(lattice)} stride {tuple(s)} taps {taps[0][0]}x{taps[1][0]}x{taps[2][0]}"):
        L.check(lib.rehr_wgrad_f32(C.byref(d), _stream()), "rehr_wgrad_f32")


def _direct_desc(x, w, bias, y, stride, pad, act, slope, stats, stats_mode):
    d = L.DirectConvDesc()
    N, Cin, Di, Hi, Wi = x.shape
    Cout, _, KD, KH, KW = w.shape
    _, _, Do, Ho, Wo = y.shape
    d.x, d.ldx, d.N, d.Di, d.Hi, d.Wi, d.Cin = _ptr(x), Cin, N, Di, Hi, Wi, Cin
    d.w, d.bias = _ptr(w), _ptr(bias)
    d.y, d.ldy, d.Do, d.Ho, d.Wo, d.Cout = _ptr(y), Cout, Do, Ho, Wo, Cout
    d.KD, d.KH, d.KW = KD, KH, KW
    d.sd, d.sh, d.sw = stride
    d.pd, d.ph, d.pw = pad
    d.act, d.slope = act, slope
    d.stats, d.stats_mode = _ptr(stats), stats_mode
    return d


def small_cin_bf16_out_ok(w_shape, stride):
    """Shapes whose thin-input forward runs on the matrix cores (thin_cin_conv.hip) and can therefore store bf16."""
    return w_shape[1] <= 2 and w_shape[0] in (32, 64) and w_shape[4] <= 8 and stride[2] in (1, 2)


def small_cin_fwd(x, w, bias, y, stride, pad, act, slope, stats, stats_mode):
    """x, w, bias fp32; y fp32, or bf16 on the mixed-precision path (matrix-core shapes only)."""
    _chk_dev(x, w, bias, y, f64=(stats,))
    if x.dtype != torch.float32:
        raise L.RehrsegHipError("thin-input conv: the input stays float32")
    d = _direct_desc(x, w.contiguous(), bias, y, stride, pad, act, slope, stats, stats_mode)
    if y.dtype == torch.bfloat16:
        L.check(L.load().rehr_conv_small_cin_fwd_ybf16(C.byref(d), _stream()), "rehr_conv_small_cin_fwd_ybf16")
        return
    L.check(L.load().rehr_conv_small_cin_fwd_f32(C.byref(d), _stream()), "rehr_conv_small_cin_fwd_f32")


def im2col(x, w, out_dims, stride, pad, Kpad):
    """(N, Kpad, Do, Ho, Wo) NDHWC columns of a thin-input conv (k = ci*T + tap)."""
    _chk_dev(x, w)
    N = x.shape[0]
    col = new_act(N, Kpad, *out_dims, like=x)
    d = _direct_desc(x, w, None, col, stride, pad, 0, 0.0, None, 0)
    d.Cout, d.ldy = w.shape[0], Kpad
    L.check(L.load().rehr_im2col_f32(C.byref(d), _ptr(col), Kpad, _stream()), "rehr_im2col_f32")
    return col


def small_cin_wgrad_on_mfma(x, w, dy, stride, pad):
    """True when the thin-input weight gradient has its own matrix-core kernel for this shape (no im2col route)."""
    d = _direct_desc(x, w.contiguous(), None, dy, stride, pad, 0, 0.0, None, 0)
    return bool(L.load().rehr_conv_small_cin_wgrad_on_mfma(C.byref(d)))


def small_cin_wgrad(x, w, dy, stride, pad, want_bias):
    """dy fp32, or bf16 (mixed precision) where the matrix-core kernel takes the shape (else cast up first)."""
    _chk_dev(x, w, dy)
    lib = L.load()
    if dy.dtype == torch.bfloat16:
        d = _direct_desc(x, w.contiguous(), None, dy, stride, pad, 0, 0.0, None, 0)
        if not lib.rehr_conv_small_cin_wgrad_on_mfma(C.byref(d)):
            dy = dy.float()
    d = _direct_desc(x, w.contiguous(), None, dy, stride, pad, 0, 0.0, None, 0)
    nbytes = lib.rehr_conv_small_cin_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty_like(w, memory_format=torch.contiguous_format)
    db = torch.empty(w.shape[0], dtype=torch.float32, device=x.device) if want_bias else None
    fn = lib.rehr_conv_small_cin_wgrad_dybf16 if dy.dtype == torch.bfloat16 else lib.rehr_conv_small_cin_wgrad_f32
    L.check(fn(C.byref(d), _ptr(dw), _ptr(db), _ptr(ws), nbytes, _stream()), "rehr_conv_small_cin_wgrad")
    return dw, db


def small_cout_fwd(x, w, bias, y, pad, act, slope):
    _chk_dev(x, w, bias, y)
    d = _direct_desc(x, w.contiguous(), bias, y, (1, 1, 1), pad, act, slope, None, 0)
    L.check(L.load().rehr_conv_small_cout_fwd_f32(C.byref(d), _stream()), "rehr_conv_small_cout_fwd_f32")


def small_cout_dgrad(dy, w, x_shape, pad):
    _chk_dev(dy, w)
    dx = new_act(*x_shape, like=dy)
    d = _direct_desc(dx, w.contiguous(), None, dy, (1, 1, 1), pad, 0, 0.0, None, 0)
    L.check(L.load().rehr_conv_small_cout_dgrad_f32(C.byref(d), _ptr(dx), _stream()),
            "rehr_conv_small_cout_dgrad_f32")
    return dx


def small_cout_wgrad(x, w, dy, pad, want_bias):
    _chk_dev(x, w, dy)
    d = _direct_desc(x, w.contiguous(), None, dy, (1, 1, 1), pad, 0, 0.0, None, 0)
    lib = L.load()
    nbytes = lib.rehr_conv_small_cout_wgrad_workspace_bytes(C.byref(d))
    if nbytes < 0:
        L.check(int(nbytes), "rehr_conv_small_cout_wgrad_workspace_bytes")
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty_like(w, memory_format=torch.contiguous_format)
    db = torch.empty(w.shape[0], dtype=torch.float32, device=x.device) if want_bias else None
    L.check(lib.rehr_conv_small_cout_wgrad_f32(C.byref(d), _ptr(dw), _ptr(db), _ptr(ws), nbytes, _stream()),
            "rehr_conv_small_cout_wgrad_f32")
    return dw, db


def se_gate_fwd(stats, w, b, N, Cc, S):
    _chk_dev(w, b, f64=(stats,))
    gate = torch.empty((N, Cc), dtype=torch.float32, device=stats.device)
    mean = torch.empty((N, Cc), dtype=torch.float32, device=stats.device)
    L.check(L.load().rehr_se_gate_fwd_f32(_ptr(stats), _ptr(w.contiguous()), _ptr(b.contiguous()), _ptr(gate),
                                          _ptr(mean), N, Cc, S, _stream()), "rehr_se_gate_fwd_f32")
    return gate, mean


def _nsc(x):
    N, Cc, D, H, W = x.shape
    return N, D * H * W, Cc


def scale_res_act_fwd(x, gate, res, act, slope):
    _chk_dev(x, gate, res)
    _same_dtype(x, res)
    N, S, Cc = _nsc(x)
    y = new_act(*x.shape, like=x)
    fn, name = _fn("rehr_scale_res_act_fwd", x)
    L.check(fn(_ptr(x), Cc, _ptr(gate), _ptr(res), Cc, _ptr(y), Cc, N, S, Cc, act, slope, _stream()), name)
    return y


def scale_res_act_bwd(dy, y, x, gate, want_dres, act, slope):
    _chk_dev(dy, y, x, gate)
    _same_dtype(dy, y, x)
    N, S, Cc = _nsc(x)
    dx = new_act(*x.shape, like=x)
    dres = new_act(*x.shape, like=x) if want_dres else None
    dgate = zeros_f64((N, Cc), x.device)
    fn, name = _fn("rehr_scale_res_act_bwd", x)
    L.check(fn(_ptr(dy), Cc, _ptr(y), Cc, _ptr(x), Cc, _ptr(gate), _ptr(dx), Cc, _ptr(dres), Cc, _ptr(dgate), N, S, Cc,
               act, slope, _stream()), name)
    return dx, dres, dgate


def se_gate_bwd(dgate, gate, mean, w, S):
    _chk_dev(gate, mean, w, f64=(dgate,))
    N, Cc = gate.shape
    dw = torch.empty((Cc, Cc), dtype=torch.float32, device=gate.device)
    db = torch.empty((Cc,), dtype=torch.float32, device=gate.device)
    k = torch.empty((N, Cc), dtype=torch.float32, device=gate.device)
    L.check(L.load().rehr_se_gate_bwd_f32(_ptr(dgate), _ptr(gate), _ptr(mean), _ptr(w.contiguous()), _ptr(dw),
                                          _ptr(db), _ptr(k), N, Cc, S, _stream()), "rehr_se_gate_bwd_f32")
    return dw, db, k


def add_channel_const(x, k):
    _chk_dev(x, k)
    N, S, Cc = _nsc(x)
    fn, name = _fn("rehr_add_channel_const", x)
    L.check(fn(_ptr(x), Cc, _ptr(k), N, S, Cc, _stream()), name)


def instnorm_act_fwd(x, stats, gamma, beta, eps, act, slope):
    _chk_dev(x, gamma, beta, f64=(stats,))
    N, S, Cc = _nsc(x)
    y = new_act(*x.shape, like=x)
    mr = torch.empty((N, Cc, 2), dtype=torch.float32, device=x.device)
    fn, name = _fn("rehr_instnorm_act_fwd", x)
    L.check(fn(_ptr(x), Cc, _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(y), Cc, _ptr(mr), N, S, Cc, eps, act, slope,
               _stream()), name)
    return y, mr


def instnorm_act_bwd(dy, x, mr, gamma, beta, act, slope, want_conv_bias=False):
    """-> (dx, dgamma, dbeta[, dconv_bias]).  want_conv_bias (bf16 only): the column sums of dx -- the gradient of the
    bias of the convolution in front -- formed inside the apply pass instead of by a separate pass over dx."""
    _chk_dev(dy, x, mr, gamma, beta)
    _same_dtype(dy, x)
    N, S, Cc = _nsc(x)
    dx = new_act(*x.shape, like=x)
    dg = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    db = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    red = zeros_f64((N, Cc, 2), x.device)
    if want_conv_bias:
        if x.dtype != torch.bfloat16:
            raise L.RehrsegHipError("the fused conv-bias gradient exists for the bf16 path (fp32: the weight-gradient kernels carry it)")
        dsum = torch.empty((Cc,), dtype=torch.float64, device=x.device)
        dcb = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        L.check(L.load().rehr_instnorm_act_bwd_dbias_bf16(_ptr(dy), Cc, _ptr(x), Cc, _ptr(mr), _ptr(gamma), _ptr(beta),
                                                          _ptr(dx), Cc, _ptr(dg), _ptr(db), _ptr(red), N, S, Cc, act,
                                                          slope, _ptr(dsum), _ptr(dcb), _stream()),
                "rehr_instnorm_act_bwd_dbias_bf16")
        return dx, dg, db, dcb
    fn, name = _fn("rehr_instnorm_act_bwd", x)
    L.check(fn(_ptr(dy), Cc, _ptr(x), Cc, _ptr(mr), _ptr(gamma), _ptr(beta), _ptr(dx), Cc, _ptr(dg), _ptr(db), _ptr(red),
               N, S, Cc, act, slope, _stream()), name)
    return dx, dg, db


def upsample_depth_fwd(x, Do):
    _chk_dev(x)
    N, Cc, Di, H, W = x.shape
    y = new_act(N, Cc, Do, H, W, like=x)
    L.check(L.load().rehr_upsample_depth_fwd_f32(_ptr(x), _ptr(y), N, Di, Do, H * W, Cc, _stream()),
            "rehr_upsample_depth_fwd_f32")
    return y


def upsample_depth_bwd(dy, Di):
    _chk_dev(dy)
    N, Cc, Do, H, W = dy.shape
    dx = new_act(N, Cc, Di, H, W, like=dy)
    L.check(L.load().rehr_upsample_depth_bwd_f32(_ptr(dy), _ptr(dx), N, Di, Do, H * W, Cc, _stream()),
            "rehr_upsample_depth_bwd_f32")
    return dx


def upmix_depth_fwd(g, bias, Do, Cc, KD, pd, act, slope):
    """g (N, KD*Cc, Di, H, W) NDHWC -> y (N, Cc, Do, H, W): depth interpolation + depth-tap sum + bias + act."""
    _chk_dev(g, bias)
    N, _, Di, H, W = g.shape
    y = new_act(N, Cc, Do, H, W, like=g)
    fn, name = _fn("rehr_upmix_depth_fwd", g)
    L.check(fn(_ptr(g), _ptr(bias), _ptr(y), N, Di, Do, H * W, Cc, KD, pd, act, slope, _stream()), name)
    return y


def upmix_depth_bwd(dy, y, Di, KD, pd, act, slope):
    """dg of upmix_depth_fwd from the output gradient dy and the saved output y (dz = dy * act'(y) on the fly)."""
    _chk_dev(dy, y)
    _same_dtype(dy, y)
    N, Cc, Do, H, W = dy.shape
    dg = new_act(N, KD * Cc, Di, H, W, like=dy)
    fn, name = _fn("rehr_upmix_depth_bwd", dy)
    L.check(fn(_ptr(dy), _ptr(y), _ptr(dg), N, Di, Do, H * W, Cc, KD, pd, act, slope, _stream()), name)
    return dg


def channel_sum_actgrad(dy, y, act, slope):
    _chk_dev(dy, y)
    _same_dtype(dy, y)
    N, S, Cc = _nsc(dy)
    out = torch.empty((Cc,), dtype=torch.float32, device=dy.device)
    scratch = torch.empty((Cc,), dtype=torch.float64, device=dy.device)
    fn, name = _fn("rehr_channel_sum_actgrad", dy)
    L.check(fn(_ptr(dy), _ptr(y), Cc, N * S, Cc, act, slope, _ptr(out), _ptr(scratch), _stream()), name)
    return out


def window_stem_assemble(g, mean, bias, B, nwin, act, slope, out_dtype=None):
    """g = three (B*(nwin+3)+1, C, 1, h, w) NDHWC fp32 per-tap responses -> y (B*nwin, C, 4, h, w) (see the header);
    out_dtype bf16: stored for a mixed-precision first block."""
    _chk_dev(*g, mean, bias)
    S, Cc, _, h, w = g[0].shape
    if S != B * (nwin + 3) + 1:
        raise ValueError("window_stem_assemble: slice count")
    if any(t.dtype != torch.float32 for t in g):
        raise L.RehrsegHipError("window_stem_assemble: the per-tap responses are float32")
    bf = out_dtype == torch.bfloat16
    y = new_act(B * nwin, Cc, 4, h, w, like=g[0], dtype=torch.bfloat16 if bf else torch.float32)
    fn = L.load().rehr_window_stem_assemble_bf16 if bf else L.load().rehr_window_stem_assemble_f32
    L.check(fn(_ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(mean), _ptr(bias), _ptr(y), B, nwin, nwin + 3, h * w, Cc, act,
               slope, _stream()), "rehr_window_stem_assemble")
    return y


def cosdist_stats(x1, x2):
    """(N, 64, D, H, W) NDHWC pair -> stats (N, 64, 3) fp64 = (S12, S11, S22) of the channel-normalised tensors."""
    _chk_dev(x1, x2)
    N, Cc, D, H, W = x1.shape
    stats = torch.empty((N, Cc, 3), dtype=torch.float64, device=x1.device)
    L.check(L.load().rehr_cosdist_stats_f32(_ptr(x1), _ptr(x2), _ptr(stats), N, D * H * W, Cc, _stream()),
            "rehr_cosdist_stats_f32")
    return stats


def cosdist_bwd(x1, x2, stats, scale):
    _chk_dev(x1, x2, f64=(stats,))
    N, Cc, D, H, W = x1.shape
    dx = new_act(N, Cc, D, H, W, like=x1)
    L.check(L.load().rehr_cosdist_bwd_f32(_ptr(x1), _ptr(x2), _ptr(stats), _ptr(dx), N, D * H * W, Cc, float(scale),
                                          _stream()), "rehr_cosdist_bwd_f32")
    return dx


def _uasr_args(om, ue, wu, bu, D):
    _chk_dev(om, ue, wu, bu)
    for t in (om, ue, wu, bu):
        if t.dtype != torch.float32:
            raise L.RehrsegHipError("uasr_mix: float32 operands")
    N, C2, one, H, W = om.shape
    K = ue.shape[1] // D
    if one != 1 or ue.shape != (N, K * D, 1, H, W) or C2 != 2 * K * D or wu.numel() != K or bu.numel() != 1:
        raise ValueError(f"uasr_mix: om {tuple(om.shape)} / ue {tuple(ue.shape)} / D {D} / wu {tuple(wu.shape)}")
    if K not in (4, 8, 16, 32):
        raise NotImplementedError(f"uasr_mix: K = {K} candidates per slice (kernels: 4, 8, 16, 32)")
    return N, K, H, W


def uasr_mix_fwd(om, ue, wu, bu, D):
    """om (N, D*2K, 1, H, W), ue (N, D*K, 1, H, W) NDHWC -> out (N, 2, D, H, W), unc (N, 1, D, H, W) (see the header)."""
    N, K, H, W = _uasr_args(om, ue, wu, bu, D)
    out = torch.empty((N, 2, D, H, W), dtype=torch.float32, device=om.device)
    unc = torch.empty((N, 1, D, H, W), dtype=torch.float32, device=om.device)
    L.check(L.load().rehr_uasr_mix_fwd_f32(_ptr(om), _ptr(ue), _ptr(wu), _ptr(bu), _ptr(out), _ptr(unc), N, K, D, H * W,
                                           _stream()), "rehr_uasr_mix_fwd_f32")
    return out, unc


def uasr_mix_bwd(om, ue, wu, bu, gout, gunc, D):
    """Gradients of uasr_mix_fwd: (dom, due, dwu (K,), dbu (1,)); gout / gunc dense NCDHW."""
    N, K, H, W = _uasr_args(om, ue, wu, bu, D)
    _chk_dev(gout, gunc)
    if gout.shape != (N, 2, D, H, W) or gunc.shape != (N, 1, D, H, W) or not (gout.is_contiguous() and gunc.is_contiguous()):
        raise ValueError("uasr_mix_bwd: gout (N,2,D,H,W) / gunc (N,1,D,H,W), contiguous")
    dom = new_act(N, 2 * K * D, 1, H, W, like=om)
    due = new_act(N, K * D, 1, H, W, like=ue)
    lib = L.load()
    blocks = lib.rehr_uasr_mix_blocks(N, D, H * W)
    partial = torch.empty((blocks, K + 1), dtype=torch.float64, device=om.device)
    L.check(lib.rehr_uasr_mix_bwd_f32(_ptr(om), _ptr(ue), _ptr(wu), _ptr(bu), _ptr(gout), _ptr(gunc), _ptr(dom), _ptr(due),
                                      _ptr(partial), N, K, D, H * W, _stream()), "rehr_uasr_mix_bwd_f32")
    tot = partial.sum(0).float()
    return dom, due, tot[:K], tot[K:]


def quad_maxpool_fwd(x):
    """x (N, C, D, H, W) NDHWC, H and W even -> (y (N*D, C, 2, 2) fp32 NHWC-dense, idx int32 same layout)."""
    _chk_dev(x)
    N, Cc, D, H, W = x.shape
    y = torch.empty((N * D, Cc, 2, 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    idx = torch.empty((N * D, 2, 2, Cc), dtype=torch.int32, device=x.device)
    L.check(L.load().rehr_quad_maxpool_fwd_f32(_ptr(x), _ptr(y), C.c_void_p(idx.data_ptr()), N * D, H, W, Cc, _stream()),
            "rehr_quad_maxpool_fwd_f32")
    return y, idx


def quad_maxpool_bwd(dy, idx, shape):
    _chk_dev(dy)
    N, Cc, D, H, W = shape
    dy = dy.contiguous(memory_format=torch.channels_last)
    dx = new_act(N, Cc, D, H, W, like=dy)
    L.check(L.load().rehr_quad_maxpool_bwd_f32(_ptr(dy), C.c_void_p(idx.data_ptr()), _ptr(dx), N * D, H, W, Cc, _stream()),
            "rehr_quad_maxpool_bwd_f32")
    return dx


def act_fwd(x, act, slope):
    _chk_dev(x)
    y = torch.empty_like(x)
    L.check(L.load().rehr_act_fwd_f32(_ptr(x), _ptr(y), x.numel(), act, slope, _stream()), "rehr_act_fwd_f32")
    return y


def act_bwd(dy, y, act, slope):
    _chk_dev(dy, y)
    _same_dtype(dy, y)
    dx = torch.empty_like(y)
    fn, name = _fn("rehr_act_bwd", y)
    L.check(fn(_ptr(dy), _ptr(y), _ptr(dx), y.numel(), act, slope, _stream()), name)
    return dx


def channel_sum(x):
    _chk_dev(x)
    N, S, Cc = _nsc(x)
    out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    scratch = torch.empty((Cc,), dtype=torch.float64, device=x.device)
    fn, name = _fn("rehr_channel_sum", x)
    L.check(fn(_ptr(x), Cc, N * S, Cc, _ptr(out), 0, _ptr(scratch), _stream()), name)
    return out


def seg_loss_fwd(logits, target, unc):
    """logits (N,C,D,H,W) NDHWC, target/unc (N,S) float32 -> stats double[N*C*3 + 1]."""
    _chk_dev(logits, target, unc)
    N, Cc = logits.shape[0], logits.shape[1]
    S = logits.shape[2] * logits.shape[3] * logits.shape[4]
    stats = torch.empty(N * Cc * 3 + 1, dtype=torch.float64, device=logits.device)
    L.check(L.load().rehr_seg_loss_fwd_f32(_ptr(logits), Cc, _ptr(target), _ptr(unc), N, Cc, S, _ptr(stats),
                                           _stream()), "rehr_seg_loss_fwd_f32")
    return stats


def seg_loss_bwd(logits, target, unc, stats, w_ce, w_dice, smooth, do_bg, grad_out):
    _chk_dev(logits, target, unc, grad_out, f64=(stats,))
    N, Cc, D, H, W = logits.shape
    dl = new_act(N, Cc, D, H, W, like=logits)
    L.check(L.load().rehr_seg_loss_bwd_f32(_ptr(logits), Cc, _ptr(target), _ptr(unc), N, Cc, D * H * W, _ptr(stats),
                                           float(w_ce), float(w_dice), float(smooth), int(do_bg), _ptr(grad_out),
                                           _ptr(dl), Cc, _stream()), "rehr_seg_loss_bwd_f32")
    return dl


def bce_dice_fwd(x, t):
    """x, t dense (N, C, S) float32 -> stats (C, 4) double = {sum bce, sum p t, sum p^2, sum t^2}."""
    _chk_dev(x, t)
    N, Cc, S = x.shape
    stats = torch.empty((Cc, 4), dtype=torch.float64, device=x.device)
    L.check(L.load().rehr_bce_dice_fwd_f32(_ptr(x), _ptr(t), N, Cc, S, _ptr(stats), _stream()), "rehr_bce_dice_fwd_f32")
    return stats


def bce_dice_bwd(x, t, stats, alpha, beta, grad_out):
    _chk_dev(x, t, grad_out, f64=(stats,))
    N, Cc, S = x.shape
    dx = torch.empty_like(x)
    L.check(L.load().rehr_bce_dice_bwd_f32(_ptr(x), _ptr(t), N, Cc, S, _ptr(stats), float(alpha), float(beta), _ptr(grad_out),
                                           _ptr(dx), _stream()), "rehr_bce_dice_bwd_f32")
    return dx


# ----------------------------------------------------------------------------- training-patch feed (section 8 f-4)
def patch_gather(items, dims, scale=1.0, bias=0.0):
    """One launch cuts len(items) patches out of device-resident volumes (rehr_patch_gather).

    items: (src, base, stride[4], lo[4], hi[4]) per patch; src is a contiguous float32 / uint8 device tensor, base and
    stride count source elements.  Returns float32 [len(items), *dims].  Every address the valid box [lo, hi) can
    produce is checked here against the volume (the kernel trusts its descriptors)."""
    dims = tuple(int(v) for v in dims)
    if len(dims) != 4 or not items:
        raise L.RehrsegHipError("patch_gather: four output axes and at least one item")
    dt = items[0][0].dtype
    if dt not in (torch.float32, torch.uint8):
        raise L.RehrsegHipError(f"patch_gather: float32 or uint8 volumes, got {dt}")
    arr = (L.PatchItem * len(items))()
    for i, (src, base, stride, lo, hi) in enumerate(items):
        if not src.is_cuda or not src.is_contiguous() or src.dtype != dt:
            raise L.RehrsegHipError("patch_gather: contiguous device volumes of one dtype (no CPU fallback)")
        lo_off = hi_off = int(base)
        empty = False
        for k in range(4):
            if not (0 <= lo[k] <= hi[k] <= dims[k]):
                raise L.RehrsegHipError(f"patch_gather: valid box {lo}..{hi} outside the patch {dims}")
            if lo[k] == hi[k]:
                empty = True
                continue
            a, b = int(stride[k]) * int(lo[k]), int(stride[k]) * (int(hi[k]) - 1)
            lo_off += min(a, b)
            hi_off += max(a, b)
        if not empty and (lo_off < 0 or hi_off >= src.numel()):
            raise L.RehrsegHipError(f"patch_gather: item {i} addresses [{lo_off}, {hi_off}] of a {src.numel()}-element volume")
        arr[i].src = src.data_ptr()
        arr[i].base = int(base)
        for k in range(4):
            arr[i].stride[k] = int(stride[k])
            arr[i].lo[k] = int(lo[k])
            arr[i].hi[k] = int(hi[k])
    out = torch.empty((len(items),) + dims, device=items[0][0].device, dtype=torch.float32)
    d = L.PatchGatherDesc()
    d.n_items = len(items)
    for k in range(4):
        d.dims[k] = dims[k]
    d.src_dtype = L.PATCH_U8 if dt == torch.uint8 else L.PATCH_F32
    d.scale, d.bias = float(scale), float(bias)
    d.dst = out.data_ptr()
    d.dst_item_stride = out[0].numel()
    L.check(L.load().rehr_patch_gather(C.byref(d), arr, _stream()), "rehr_patch_gather")
    return out


def axis_resample(x, axis, idx, w, validated=False):
    """y[..., j, ...] = sum_t w[j, t] * x[..., idx[j, t], ...] along `axis` (idx < 0 drops the term); x contiguous float32
    on the device, idx int32 [n_out, taps], w float32 [n_out, taps] (device).  `validated`: the caller has already
    checked idx < x.shape[axis] on the host (saves the device round trip of the check here)."""
    _chk_dev(x, w)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise L.RehrsegHipError("axis_resample: contiguous float32 input")
    if idx.dtype != torch.int32 or idx.shape != w.shape or idx.dim() != 2 or not idx.is_cuda:
        raise L.RehrsegHipError("axis_resample: idx int32 / w float32 tables of one [n_out, taps] shape on the device")
    axis = axis % x.dim()
    n_in = x.shape[axis]
    if not validated and idx.numel() and int(idx.max()) >= n_in:
        raise L.RehrsegHipError("axis_resample: tap index beyond the axis")
    outer = 1
    for s in x.shape[:axis]:
        outer *= s
    inner = 1
    for s in x.shape[axis + 1:]:
        inner *= s
    n_out, taps = idx.shape
    y = torch.empty(x.shape[:axis] + (n_out,) + x.shape[axis + 1:], device=x.device, dtype=torch.float32)
    if y.numel() == 0:
        return y
    L.check(L.load().rehr_axis_resample_f32(_ptr(x), _ptr(y), _ptr(idx.contiguous()), _ptr(w.contiguous()), outer, n_in,
                                            n_out, inner, taps, _stream()), "rehr_axis_resample_f32")
    return y


# ----------------------------------------------------------------------------- sr_head.2 on the bf16 matrix cores
def _thin5_ws(d, dev, f32=False):
    fn = L.load().rehr_conv5_thin_f32_workspace_bytes if f32 else L.load().rehr_conv5_thin_workspace_bytes
    n = int(fn(C.byref(d)))
    if n < 0:
        L.check(n, "rehr_conv5_thin_workspace_bytes")
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=dev), n


USE_THIN5_F32 = True   # sr_head.2 of the fp32 path on the fp32 matrix cores (False: the VALU kernels of direct_conv.hip)


def thin5_supported(x_shape, w_shape, pad, dtype=torch.bfloat16):
    """Conv3d(16 -> 2, 5x5x5, stride 1, pad 2) with W % 32 == 0 and W <= 160 (bf16) / 128 (fp32): the shapes the
    matrix-core kernels of thin_conv_{bf16,f32}.hip take."""
    N, Cin, D, H, W = x_shape
    if dtype == torch.float32 and not USE_THIN5_F32:
        return False
    es, wmax = (2, 160) if dtype == torch.bfloat16 else (4, 128)
    return (tuple(w_shape) == (2, 16, 5, 5, 5) and Cin == 16 and tuple(pad) == (2, 2, 2) and W % 32 == 0
            and 32 <= W <= wmax and D * H * W * 16 * es < 2 ** 32 and dtype in (torch.bfloat16, torch.float32))


def thin5_fwd(x, w, bias):
    """y (fp32) = conv3d(x (bf16 or fp32 NDHWC), w, bias) for sr_head.2 (rehr_conv5_thin_fwd_{bf16,f32})."""
    _chk_dev(x, w, bias)
    N, Cin, D, H, W = x.shape
    y = new_act(N, 2, D, H, W, like=x, dtype=torch.float32)
    d = _direct_desc(x, w.contiguous(), bias, y, (1, 1, 1), (2, 2, 2), 0, 0.0, None, 0)
    f32 = x.dtype == torch.float32
    ws, n = _thin5_ws(d, x.device, f32)
    fn, name = _fn("rehr_conv5_thin_fwd", x)
    L.check(fn(C.byref(d), _ptr(ws), n, _stream()), name)
    return y


def thin5_dgrad(dy, w, dtype=torch.bfloat16):
    """dx (NDHWC, 16 channels, `dtype`) of sr_head.2 from dY (fp32 NDHWC, 2 channels)."""
    _chk_dev(dy, w)
    if dy.dtype != torch.float32:
        raise L.RehrsegHipError("thin5_dgrad: fp32 output gradient")
    N, _, D, H, W = dy.shape
    dx = new_act(N, 16, D, H, W, like=dy, dtype=dtype)
    d = _direct_desc(dx, w.contiguous(), None, dy, (1, 1, 1), (2, 2, 2), 0, 0.0, None, 0)
    ws, n = _thin5_ws(d, dy.device, dtype == torch.float32)
    fn, name = _fn("rehr_conv5_thin_dgrad", dx)
    L.check(fn(C.byref(d), _ptr(dx), 16, _ptr(ws), n, _stream()), name)
    return dx


def thin5_wgrad(x, w, dy, want_bias=False):
    """(dw (2,16,5,5,5), db) fp32 of sr_head.2 from x (bf16 or fp32) and dY (fp32)."""
    _chk_dev(x, w, dy)
    if dy.dtype != torch.float32:
        raise L.RehrsegHipError("thin5_wgrad: fp32 output gradient")
    d = _direct_desc(x, w.contiguous(), None, dy, (1, 1, 1), (2, 2, 2), 0, 0.0, None, 0)
    ws, n = _thin5_ws(d, x.device, x.dtype == torch.float32)
    dw = torch.empty_like(w, memory_format=torch.contiguous_format)
    db = torch.empty(2, dtype=torch.float32, device=x.device) if want_bias else None
    fn, name = _fn("rehr_conv5_thin_wgrad", x)
    L.check(fn(C.byref(d), _ptr(dw), _ptr(db), _ptr(ws), n, _stream()), name)
    return dw, db
