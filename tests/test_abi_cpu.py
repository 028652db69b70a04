"""The C-ABI library loads without a GPU and exports exactly what include/rehrseg_hip.h
declares; argument validation rejects malformed descriptors before any launch."""
import ctypes as C

import pytest

from rehrseg_amd import lib as L


def test_library_exports_every_declared_symbol():
    lib = L.load()
    declared = L.declared_symbols()
    assert len(declared) >= 20
    assert set(declared) == set(L.PROTOTYPES), set(declared) ^ set(L.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rehr_abi_version() == L.ABI_VERSION


def test_struct_sizes_match_header_layout():
    # int32/float/pointer fields only, natural alignment: sizes are a cheap layout check
    assert C.sizeof(L.AxisTaps) == 20
    assert C.sizeof(L.GatherGemmDesc) % 8 == 0 and C.sizeof(L.WgradDesc) % 8 == 0


def test_malformed_descriptors_are_rejected_without_launching():
    lib = L.load()
    d = L.GatherGemmDesc()  # all zero: null pointers
    assert lib.rehr_gather_gemm_f32(C.byref(d), None) == -1
    w = L.WgradDesc()
    assert lib.rehr_wgrad_f32(C.byref(w), None) == -1
    assert lib.rehr_wgrad_workspace_bytes(C.byref(w)) == -1
    assert lib.rehr_pack_weights_f32(None, None, 1, 1, 1, 1, 0, None) == -1
    assert lib.rehr_act_fwd_f32(None, None, 4, 0, 0.0, None) == -1
    dc = L.DirectConvDesc()
    assert lib.rehr_conv_small_cin_fwd_f32(C.byref(dc), None) == -1


def test_product_path_refuses_cpu_tensors():
    import torch
    from rehrseg_amd import ops
    with pytest.raises(L.RehrsegHipError):
        ops.fused_conv3d(torch.randn(1, 32, 2, 4, 4), torch.randn(32, 32, 3, 3, 3), None, 1, 1)


def test_flag_constants_of_the_binding_equal_the_header():
    """Every REHR_GG_* / REHR_DBG_* bit the Python binding uses carries the value the header defines (a renumbered bit
    would silently select another kernel or, for REHR_GG_WS_ONLY / _READY, skip or repeat a weight transform)."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rehrseg_hip.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+REHR_(\w+)\s+(0x[0-9a-fA-F]+|\d+)\s", hdr)}
    checked = 0
    for name in dir(L):
        if name.startswith(("GG_", "DBG_")) and isinstance(getattr(L, name), int):
            assert name in defs, f"lib.{name} has no REHR_{name} in the header"
            assert defs[name] == getattr(L, name), (name, defs[name], getattr(L, name))
            checked += 1
    assert checked >= 12
    assert defs["GG_WS_READY"] != defs["GG_WS_ONLY"] and defs["GG_Y_F32"] == 1
