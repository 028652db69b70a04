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
