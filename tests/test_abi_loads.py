"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/licv_hip.h
declares, and the product refuses to run without it / without a GPU (no silent fallback)."""
import ctypes

import pytest
import torch

from licv import _lib


def test_library_exports_every_declared_symbol():
    names = _lib.declared_symbols()
    assert {"licv_gemm_bf16", "licv_attn_fwd", "licv_inject_renorm_fwd", "licv_inject_renorm_bwd",
            "licv_rmsnorm_fwd", "licv_rotary_fwd", "licv_layernorm_fwd"} <= set(names)
    assert _lib.check_exports()
    assert _lib.lib().licv_version() == 1


def test_bad_arguments_are_reported_not_crashed():
    l = _lib.lib()
    # null pointers / bad sizes are rejected on the host before any launch
    st = l.licv_inject_renorm_fwd(None, 0, None, None, None, 4, 64, None, None, ctypes.c_float(1e-6), None)
    assert st == -1 and b"null pointer" in l.licv_last_error()
    ep = _lib.GemmEpilogue()
    st = l.licv_gemm_bf16(ctypes.c_void_p(16), 12, ctypes.c_void_p(16), 12, ctypes.c_void_p(16), 16, 4, 16, 12,
                          ctypes.byref(ep), None)
    assert st == -1 and b"multiples of 8" in l.licv_last_error()


def test_missing_library_is_a_hard_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.LicvError, match="no CPU fallback"):
        _lib.lib()


def test_ops_refuse_cpu_tensors():
    from licv import ops
    with pytest.raises(AssertionError, match="device memory only"):
        ops.inject_renorm(torch.zeros(2, 8), torch.zeros(8))
