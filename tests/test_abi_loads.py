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
    # the binding, the header and the built library agree on the ABI version (bumped with every change of include/licv_hip.h)
    assert _lib.lib().licv_version() == _lib.ABI_VERSION == _lib.header_abi_version() >= 3


def test_version_mismatch_is_refused(monkeypatch):
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.LicvError, match="ABI version"):
        _lib.check_exports()


def test_lab_library_is_separate_and_the_product_refuses_its_kernels_without_it():
    """Measurement code (experiment GEMM kernels, roofline probes) lives in liblicv_hip_lab.so: the product library exports none of it,
    and a licv_gemm_select() value naming a lab kernel is an error until the lab library has registered itself."""
    l = _lib.lib()
    for name in _lib.declared_symbols(_lib.LAB_HEADER):
        assert not hasattr(l, name), f"{name} is measurement code and must not be exported by liblicv_hip.so"
    assert not (set(_lib.declared_symbols()) & set(_lib.declared_symbols(_lib.LAB_HEADER)))
    import subprocess, sys
    # in a fresh process (this one may already have loaded the lab library): no lab -> LICV_E_UNSUPPORTED before any launch
    code = ("import ctypes, sys; sys.path[:0] = %r; from licv import _lib; l = _lib.lib(); l.licv_gemm_select(6); ep = _lib.GemmEpilogue(); "
            "st = l.licv_gemm_bf16(ctypes.c_void_p(16), 128, ctypes.c_void_p(16), 128, ctypes.c_void_p(16), 256, 512, 256, 128, ctypes.byref(ep), None); "
            "assert st == -2 and b'liblicv_hip_lab.so' in l.licv_last_error(), (st, l.licv_last_error()); "
            "assert _lib.check_lab_exports(); print('ok')") % (sys.path[:3],)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


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
