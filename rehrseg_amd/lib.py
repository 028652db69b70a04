"""ctypes binding of librehrseg_hip.so (the C-ABI declared in include/rehrseg_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
launch is rejected this module raises.  ``oracle/`` is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# REHRSEG_HIP_LIB: another build of the same ABI (kernel A/B runs, tools/ab_lib.py); default = the in-tree library
LIB_PATH = os.environ.get("REHRSEG_HIP_LIB") or os.path.join(_HERE, "librehrseg_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rehrseg_hip.h")

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
ABI_VERSION = 4
GG_Y_F32 = 1            # rehr_gather_gemm_desc.flags
GG_WS_READY = 2
GG_WS_ONLY = 4
# debug_flags (unstable; tests and A/B tools only -- 0 selects the measured-best kernels)
DBG_WGRAD_DIRECT = 1    # rehr_wgrad_desc.debug_flags
DBG_WGRAD_NO_TAP_SKIP = 2
DBG_WGRAD_NO_TAP_COLOCATE = 4
DBG_GG_NO_HALO = 1      # rehr_gather_gemm_desc.debug_flags
DBG_GG_NO_FLAT8 = 2
DBG_GG_FLAT8_HALF = 4
DBG_GG_FLAT8_FULL = 8
DBG_GG_W32P_TWO_PER_CU = 16
DBG_GG_W32P_ONE_PER_CU = 32
DBG_GG_INTERLEAVE = 64
DBG_GG_NO_TCONV_KS = 128
DBG_GG_SLICE_MAJOR = 256

_i32, _i64, _f32 = C.c_int32, C.c_int64, C.c_float
_vp = C.c_void_p


class AxisTaps(C.Structure):
    _fields_ = [("count", _i32), ("off0", _i32), ("offs", _i32), ("k0", _i32), ("ks", _i32)]

    def __init__(self, count=1, off0=0, offs=1, k0=0, ks=1):
        super().__init__(count, off0, offs, k0, ks)

    def astuple(self):
        return (self.count, self.off0, self.offs, self.k0, self.ks)


class GatherGemmDesc(C.Structure):
    _fields_ = [
        ("x1", _vp), ("x2", _vp), ("c1", _i32), ("ldx1", _i32), ("ldx2", _i32),
        ("N", _i32), ("Di", _i32), ("Hi", _i32), ("Wi", _i32), ("Cin", _i32),
        ("Ld", _i32), ("Lh", _i32), ("Lw", _i32),
        ("sd", _i32), ("sh", _i32), ("sw", _i32), ("bd", _i32), ("bh", _i32), ("bw", _i32),
        ("td", AxisTaps), ("th", AxisTaps), ("tw", AxisTaps), ("KH", _i32), ("KW", _i32),
        ("wp", _vp), ("Npad", _i32),
        ("y", _vp), ("Dy", _i32), ("Hy", _i32), ("Wy", _i32), ("Cout", _i32), ("ldy", _i32),
        ("osd", _i32), ("osh", _i32), ("osw", _i32), ("obd", _i32), ("obh", _i32), ("obw", _i32),
        ("bias", _vp), ("act", _i32), ("slope", _f32),
        ("stats", _vp), ("stats_mode", _i32),
        ("tile_d", _i32), ("tile_h", _i32), ("tile_w", _i32),
        ("wino_ws", _vp), ("wino_ws_bytes", _i64),
        ("flags", _i32), ("debug_flags", _i32),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("l", _vp), ("ldl", _i32), ("Ca", _i32),
        ("g", _vp), ("ldg", _i32), ("Cg", _i32),
        ("N", _i32), ("Ld", _i32), ("Lh", _i32), ("Lw", _i32),
        ("Dg", _i32), ("Hg", _i32), ("Wg", _i32),
        ("sd", _i32), ("sh", _i32), ("sw", _i32), ("bd", _i32), ("bh", _i32), ("bw", _i32),
        ("td", AxisTaps), ("th", AxisTaps), ("tw", AxisTaps), ("KH", _i32), ("KW", _i32),
        ("dst", _vp), ("dst_sa", _i64), ("dst_sc", _i64), ("dst_st", _i64),
        ("accumulate", _i32),
        ("workspace", _vp), ("workspace_bytes", _i64),
        ("dbias", _vp),
        ("flags", _i32), ("debug_flags", _i32),
    ]


class DirectConvDesc(C.Structure):
    _fields_ = [
        ("x", _vp), ("ldx", _i32), ("N", _i32), ("Di", _i32), ("Hi", _i32), ("Wi", _i32), ("Cin", _i32),
        ("w", _vp), ("bias", _vp),
        ("y", _vp), ("ldy", _i32), ("Do", _i32), ("Ho", _i32), ("Wo", _i32), ("Cout", _i32),
        ("KD", _i32), ("KH", _i32), ("KW", _i32), ("sd", _i32), ("sh", _i32), ("sw", _i32),
        ("pd", _i32), ("ph", _i32), ("pw", _i32),
        ("act", _i32), ("slope", _f32), ("stats", _vp), ("stats_mode", _i32),
    ]


PATCH_F32, PATCH_U8, PATCH_MAX_ITEMS = 0, 1, 16


class PatchItem(C.Structure):
    _fields_ = [("src", _vp), ("base", _i64), ("stride", _i64 * 4), ("lo", _i32 * 4), ("hi", _i32 * 4)]


class PatchGatherDesc(C.Structure):
    _fields_ = [("n_items", _i32), ("dims", _i32 * 4), ("src_dtype", _i32), ("scale", _f32), ("bias", _f32),
                ("dst", _vp), ("dst_item_stride", _i64)]


_P_GG, _P_WG, _P_DC = C.POINTER(GatherGemmDesc), C.POINTER(WgradDesc), C.POINTER(DirectConvDesc)

# name -> (restype, argtypes); must list every function include/rehrseg_hip.h declares
PROTOTYPES = {
    "rehr_gather_gemm_wino_bytes": (_i64, [_P_GG]),
    "rehr_gather_gemm_f32": (C.c_int, [_P_GG, _vp]),
    "rehr_gather_gemm_multi_f32": (C.c_int, [_P_GG, _i32, _vp]),
    "rehr_gather_gemm_bf16": (C.c_int, [_P_GG, _vp]),
    "rehr_gather_gemm_multi_bf16": (C.c_int, [_P_GG, _i32, _vp]),
    "rehr_pack_weights_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "rehr_sum_slabs_bias_act_f32": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "rehr_sum_slabs_stats_f32": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _vp, _vp]),
    "rehr_sum_slabs_bias_act_bf16": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "rehr_sum_slabs_stats_bf16": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _vp, _vp]),
    "rehr_wgrad_workspace_bytes": (_i64, [_P_WG]),
    "rehr_wgrad_uses_winograd": (C.c_int, [_P_WG]),
    "rehr_wgrad_f32": (C.c_int, [_P_WG, _vp]),
    "rehr_wgrad_bf16_workspace_bytes": (_i64, [_P_WG]),
    "rehr_wgrad_bf16": (C.c_int, [_P_WG, _vp]),
    "rehr_pack_weights_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "rehr_conv_small_cin_fwd_f32": (C.c_int, [_P_DC, _vp]),
    "rehr_im2col_f32": (C.c_int, [_P_DC, _vp, _i32, _vp]),
    "rehr_conv_small_cin_wgrad_workspace_bytes": (_i64, [_P_DC]),
    "rehr_conv_small_cin_wgrad_on_mfma": (C.c_int, [_P_DC]),
    "rehr_conv_small_cin_fwd_ybf16": (C.c_int, [_P_DC, _vp]),
    "rehr_conv_small_cin_wgrad_dybf16": (C.c_int, [_P_DC, _vp, _vp, _vp, _i64, _vp]),
    "rehr_conv_small_cin_wgrad_f32": (C.c_int, [_P_DC, _vp, _vp, _vp, _i64, _vp]),
    "rehr_conv_small_cout_fwd_f32": (C.c_int, [_P_DC, _vp]),
    "rehr_conv_small_cout_dgrad_f32": (C.c_int, [_P_DC, _vp, _vp]),
    "rehr_conv_small_cout_wgrad_workspace_bytes": (_i64, [_P_DC]),
    "rehr_conv_small_cout_wgrad_f32": (C.c_int, [_P_DC, _vp, _vp, _vp, _i64, _vp]),
    "rehr_se_gate_fwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp]),
    "rehr_scale_res_act_fwd_f32": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp]),
    "rehr_scale_res_act_bwd_f32": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp,
                                             _i32, _i64, _i32, _i32, _f32, _vp]),
    "rehr_se_gate_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp]),
    "rehr_add_channel_const_f32": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "rehr_instnorm_act_fwd_f32": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _f32,
                                            _i32, _f32, _vp]),
    "rehr_instnorm_act_bwd_f32": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                            _i32, _i64, _i32, _i32, _f32, _vp]),
    "rehr_upsample_depth_fwd_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    "rehr_upsample_depth_bwd_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    "rehr_upmix_depth_fwd_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i32, _i32, _i32, C.c_float, _vp]),
    "rehr_upmix_depth_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i32, _i32, _i32, C.c_float, _vp]),
    "rehr_channel_sum_actgrad_f32": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _i32, C.c_float, _vp, _vp, _vp]),
    "rehr_cosdist_stats_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _vp]),
    "rehr_cosdist_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, C.c_float, _vp]),
    "rehr_uasr_mix_blocks": (_i32, [_i32, _i32, _i64]),
    "rehr_uasr_mix_fwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp]),
    "rehr_uasr_mix_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp]),
    "rehr_quad_maxpool_fwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rehr_quad_maxpool_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rehr_window_stem_assemble_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i32, C.c_float, _vp]),
    "rehr_window_stem_assemble_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i32, C.c_float, _vp]),
    "rehr_conv5_thin_f32_workspace_bytes": (_i64, [_P_DC]),
    "rehr_conv5_thin_f32_supported": (C.c_int, [_P_DC]),
    "rehr_conv5_thin_fwd_f32": (C.c_int, [_P_DC, _vp, _i64, _vp]),
    "rehr_conv5_thin_dgrad_f32": (C.c_int, [_P_DC, _vp, _i32, _vp, _i64, _vp]),
    "rehr_conv5_thin_wgrad_f32": (C.c_int, [_P_DC, _vp, _vp, _vp, _i64, _vp]),
    "rehr_conv5_thin_workspace_bytes": (_i64, [_P_DC]),
    "rehr_conv5_thin_dgrad_bf16": (C.c_int, [_P_DC, _vp, _i32, _vp, _i64, _vp]),
    "rehr_conv5_thin_wgrad_bf16": (C.c_int, [_P_DC, _vp, _vp, _vp, _i64, _vp]),
    "rehr_upmix_depth_fwd_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i32, _i32, _i32, C.c_float, _vp]),
    "rehr_upmix_depth_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i32, _i32, _i32, C.c_float, _vp]),
    "rehr_channel_sum_actgrad_bf16": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _i32, C.c_float, _vp, _vp, _vp]),
    "rehr_conv5_thin_supported": (C.c_int, [_P_DC]),
    "rehr_conv5_thin_fwd_bf16": (C.c_int, [_P_DC, _vp, _i64, _vp]),
    "rehr_patch_gather": (C.c_int, [C.POINTER(PatchGatherDesc), C.POINTER(PatchItem), _vp]),
    "rehr_axis_resample_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i32, _vp]),
    "rehr_seg_loss_fwd_f32": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _vp]),
    "rehr_seg_loss_bwd_f32": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _f32, _f32, _f32, _i32, _vp,
                                        _vp, _i32, _vp]),
    "rehr_bce_dice_fwd_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp, _vp]),
    "rehr_bce_dice_bwd_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp, _f32, _f32, _vp, _vp, _vp]),
    "rehr_act_fwd_f32": (C.c_int, [_vp, _vp, _i64, _i32, _f32, _vp]),
    "rehr_act_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "rehr_channel_sum_f32": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _i32, _vp, _vp]),
    "rehr_copy_channels_f32": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "rehr_nchw_to_nhwc_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp]),
    "rehr_nhwc_to_nchw_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp]),
    "rehr_scale_res_act_fwd_bf16": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp]),
    "rehr_scale_res_act_bwd_bf16": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp,
                                              _i32, _i64, _i32, _i32, _f32, _vp]),
    "rehr_add_channel_const_bf16": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "rehr_instnorm_act_fwd_bf16": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _f32,
                                             _i32, _f32, _vp]),
    "rehr_instnorm_act_bwd_bf16": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                             _i32, _i64, _i32, _i32, _f32, _vp]),
    "rehr_instnorm_act_bwd_dbias_bf16": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                                   _i32, _i64, _i32, _i32, _f32, _vp, _vp, _vp]),
    "rehr_channel_sum_bf16": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _i32, _vp, _vp]),
    "rehr_act_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "rehr_abi_version": (C.c_int, []),
    "rehr_last_hip_error": (C.c_char_p, []),
}

_ERR = {-1: "REHR_EINVAL (malformed descriptor)", -2: "REHR_ENOSUP (unsupported shape)",
        -3: "REHR_EHIP (hip launch error)"}


class RehrsegHipError(RuntimeError):
    pass


def declared_symbols(header_path: str = HEADER_PATH):
    """Function names declared in the public header (used by the symbol test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rehr_[a-z0-9_]+)\s*\(", text)))


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RehrsegHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    # The library's code objects must register with the HIP runtime torch already
    # initialised: dlopen-ing it first leaves the process with a runtime that reports
    # "no ROCm-capable device" at the first launch (observed on the MI355X box).
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    ver = lib.rehr_abi_version()
    if ver != ABI_VERSION:
        raise RehrsegHipError(f"ABI version mismatch: library {ver}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        detail = ""
        if rc == -3 and _lib is not None:
            detail = f": {_lib.rehr_last_hip_error().decode()}"
        raise RehrsegHipError(f"{what} failed: {_ERR.get(rc, rc)}{detail}")
