"""Tensor-level wrappers over the C-ABI (torch only provides device memory and the stream)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import LICV_BF16, LICV_F32, AttnArgs, GemmEpilogue, check

# Optional launch profiler: when a list is installed, every GEMM and hook launch is bracketed by two events
# on the launch stream and (kind, event0, event1, algorithmic work) is appended; bench.py derives
# roofline.achieved from it (work = FLOPs for "gemm", bytes for "inject").
_prof = None


def set_profiler(store):
    global _prof
    _prof = store


def profiling() -> bool:
    """True while a launch profiler is installed (the per-launch event pairs need the Python-level launches)."""
    return _prof is not None


def _timed(kind, work, fn, label=None):
    if _prof is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    _prof.append((kind, e0, e1, work) if label is None else (kind, e0, e1, work, label))
    return r


ACT = {None: 0, "none": 0, "gelu": 1, "gelu_tanh": 2, "gelu_pytorch_tanh": 2, "relu": 3}


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return LICV_BF16
    if t.dtype == torch.float32:
        return LICV_F32
    raise TypeError(f"unsupported dtype {t.dtype} (bf16 or fp32 only)")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    assert t.is_cuda, "liblicv_hip operates on device memory only (no CPU fallback)"
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _bf16c(t: torch.Tensor, name: str):
    assert t.dtype == torch.bfloat16 and t.is_contiguous(), f"{name}: contiguous bf16 tensor expected"


# ------------------------------------------------------------------------------------------ hook
def inject_renorm(h: torch.Tensor, icv_row: torch.Tensor, alpha: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, norm_weight: Optional[torch.Tensor] = None,
                  norm_eps: float = 1e-6, pre: Optional[torch.Tensor] = None):
    """h: (..., H) bf16/fp32 contiguous; icv_row: (H,) fp32; alpha: 0-d/1-elem fp32 tensor or None.
    Returns fp32 h' (and the bf16 RMSNorm of h' when norm_weight is given).
    pre: bf16 branch (same shape) added to h first, in h's dtype — the layer's last residual add folded into the hook."""
    H = h.shape[-1]
    assert h.is_contiguous() and icv_row.dtype == torch.float32 and icv_row.numel() == H and icv_row.is_contiguous()
    rows = h.numel() // H
    if out is None:
        out = torch.empty(h.shape, dtype=torch.float32, device=h.device)
    xn = None
    if norm_weight is not None:
        _bf16c(norm_weight, "norm_weight")
        xn = torch.empty(h.shape, dtype=torch.bfloat16, device=h.device)
    if alpha is not None:
        assert alpha.dtype == torch.float32 and alpha.numel() == 1
    nbytes = rows * H * (h.element_size() + 4 + (2 if norm_weight is not None else 0) + (2 if pre is not None else 0))
    if pre is not None:
        assert pre.dtype == torch.bfloat16 and pre.is_contiguous() and pre.numel() == h.numel()
        _timed("inject", float(nbytes), lambda: check(_lib.lib().licv_inject_renorm_pre_fwd(
            _p(h), _dt(h), _p(pre), _p(icv_row), _p(alpha), _p(out), rows, H, _p(norm_weight), _p(xn), float(norm_eps), _stream(h))))
    else:
        _timed("inject", float(nbytes), lambda: check(_lib.lib().licv_inject_renorm_fwd(
            _p(h), _dt(h), _p(icv_row), _p(alpha), _p(out), rows, H, _p(norm_weight), _p(xn), float(norm_eps), _stream(h))))
    return (out, xn) if norm_weight is not None else out


def inject_renorm_add(branch: torch.Tensor, icv_row: torch.Tensor, residual: torch.Tensor, alpha: Optional[torch.Tensor] = None,
                      norm_weight: Optional[torch.Tensor] = None, norm_eps: float = 1e-6, norm_flavour: int = 1):
    """Hook on a residual branch: out = residual + (b+v)/||b+v||*||b|| (fp32); optional fused RMSNorm of `out`."""
    H = branch.shape[-1]
    assert branch.is_contiguous() and residual.is_contiguous() and residual.shape == branch.shape
    rows = branch.numel() // H
    out = torch.empty(branch.shape, dtype=torch.float32, device=branch.device)
    xn = torch.empty(branch.shape, dtype=torch.bfloat16, device=branch.device) if norm_weight is not None else None
    nbytes = rows * H * (branch.element_size() + residual.element_size() + 4 + (2 if norm_weight is not None else 0))
    _timed("inject", float(nbytes), lambda: check(_lib.lib().licv_inject_renorm_add_fwd(
        _p(branch), _dt(branch), _p(icv_row), _p(alpha), _p(residual), _dt(residual), _p(out), rows, H, _p(norm_weight), _p(xn),
        float(norm_eps), norm_flavour, _stream(branch))))
    return (out, xn) if norm_weight is not None else out


def scatter_rows_(out: torch.Tensor, idx: torch.Tensor, src: torch.Tensor):
    assert idx.dtype == torch.int64 and idx.is_contiguous() and src.is_contiguous() and out.is_contiguous()
    check(_lib.lib().licv_scatter_rows(_p(src), _p(idx), _p(out), idx.numel(), src.shape[-1], _stream(out)))
    return out


def inject_renorm_bwd(h: torch.Tensor, icv_row: torch.Tensor, alpha: Optional[torch.Tensor], grad_out: torch.Tensor,
                      need_grad_h: bool = True):
    """Returns (grad_h fp32 or None, grad_v (H,) fp32) with v = alpha*icv_row."""
    H = h.shape[-1]
    rows = h.numel() // H
    assert grad_out.dtype == torch.float32 and grad_out.is_contiguous() and h.is_contiguous()
    gh = torch.empty(h.shape, dtype=torch.float32, device=h.device) if need_grad_h else None
    nparts = int(_lib.lib().licv_inject_bwd_partials(rows))
    part = torch.empty((nparts, H), dtype=torch.float32, device=h.device)
    check(_lib.lib().licv_inject_renorm_bwd(_p(h), _dt(h), _p(icv_row), _p(alpha), _p(grad_out), _p(gh), _p(part),
                                            rows, H, _stream(h)))
    return gh, part.sum(0)


# ------------------------------------------------------------------------------------------ norms
def add_rmsnorm_(h: torch.Tensor, branch: torch.Tensor, w: torch.Tensor, eps: float, flavour: int = 0,
                 row_gate: Optional[torch.Tensor] = None, scale: Optional[float] = None) -> torch.Tensor:
    """h += branch (in place, h's dtype: a bf16 stream rounds the sum), returns the bf16 RMSNorm of the new h.
    row_gate (rows,) fp32 / scale: the gated cross-attention epilogue (row -> 0 where the gate is 0, then bf16(scale * branch))."""
    dim = h.shape[-1]
    rows = h.numel() // dim
    assert h.is_contiguous() and branch.is_contiguous() and branch.dtype == torch.bfloat16 and branch.numel() == h.numel()
    _bf16c(w, "w")
    if row_gate is not None:
        assert row_gate.dtype == torch.float32 and row_gate.numel() >= rows
    out = torch.empty(h.shape, dtype=torch.bfloat16, device=h.device)
    check(_lib.lib().licv_add_rmsnorm_fwd(_p(h), _dt(h), _p(branch), _p(row_gate), 0 if scale is None else 1, 0.0 if scale is None else float(scale),
                                          _p(w), _p(out), rows, dim, float(eps), flavour, _stream(h)))
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, flavour: int = 0, out: Optional[torch.Tensor] = None,
            inner: int = 1, ld_x: Optional[int] = None, ld_out: Optional[int] = None, rows: Optional[int] = None,
            dim: Optional[int] = None):
    """Dense case: x (..., dim) contiguous.  Strided per-head case: pass rows/dim/inner/ld explicitly."""
    dim = x.shape[-1] if dim is None else dim
    rows = x.numel() // dim if rows is None else rows
    ld_x = dim * inner if ld_x is None else ld_x
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    ld_out = dim * inner if ld_out is None else ld_out
    check(_lib.lib().licv_rmsnorm_fwd(_p(x), _dt(x), _p(w), _p(out), rows, dim, inner, ld_x, ld_out, float(eps), flavour, _stream(x)))
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None,
              inner: int = 1, ld_x: Optional[int] = None, ld_out: Optional[int] = None, rows: Optional[int] = None,
              dim: Optional[int] = None, out_group: int = 0, out_group_extra: int = 0):
    dim = x.shape[-1] if dim is None else dim
    rows = x.numel() // dim if rows is None else rows
    ld_x = dim * inner if ld_x is None else ld_x
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    ld_out = dim * inner if ld_out is None else ld_out
    check(_lib.lib().licv_layernorm_fwd(_p(x), _p(w), _p(b), _p(out), rows, dim, inner, ld_x, ld_out, out_group,
                                        out_group_extra, float(eps), _stream(x)))
    return out


def rotary_(buf: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, position_ids: torch.Tensor, rows: int, n_heads: int,
            head_dim: int, ld: int, tensor_stride: int, n_tensors: int):
    """In place on q (and k) inside a (rows, ld) bf16 buffer."""
    assert position_ids.dtype == torch.int64 and position_ids.is_contiguous() and position_ids.numel() == rows
    check(_lib.lib().licv_rotary_fwd(_p(buf), _p(cos), _p(sin), _p(position_ids), rows, n_heads, head_dim, ld,
                                     tensor_stride, n_tensors, cos.shape[0], _stream(buf)))
    return buf


# ------------------------------------------------------------------------------------------ GEMM
SPLITK = True               # False: never take the split-K path (A/B timing, kernel-agreement tests); set through set_splitk()


def set_splitk(on: bool) -> None:
    """Split-K on/off for every caller: this module's linear() and the library's own plan (the native layer runner asks it)."""
    global SPLITK
    SPLITK = bool(on)
    check(_lib.lib().licv_gemm_experiment(4, 1 if on else 0))


_ws: dict = {}


def _splitk_plan(M: int, N: int, K: int):
    """The library's own plan, asked on every call (host-only arithmetic): linear() and the native layer runner, which asks
    the same function, can then never disagree about the kernel of a shape — also when the library-side switch
    (`licv_gemm_experiment(4, .)`) is toggled directly."""
    sp, nb = C.c_int(1), C.c_int64(0)
    check(_lib.lib().licv_gemm_splitk_plan(M, N, K, C.byref(sp), C.byref(nb)))
    return sp.value, nb.value


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch for split-K partial tiles; launches on one stream serialise, so consecutive GEMMs may share it."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None, swiglu: bool = False,
           row_gate: Optional[torch.Tensor] = None, scale: Optional[float] = None, residual: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, M: Optional[int] = None,
           lda: Optional[int] = None, ldc: Optional[int] = None, ld_res: Optional[int] = None):
    """out = epilogue(a @ w.T).  a: (M, K) bf16 (row stride lda), w: (N, K) bf16 contiguous."""
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.is_contiguous()
    K = w.shape[1]
    N = w.shape[0]
    if M is None:
        M = a.numel() // K
    lda = K if lda is None else lda
    n_out = N // 2 if swiglu else N
    if out_dtype is None:
        out_dtype = residual.dtype if residual is not None else torch.bfloat16
    ret = None
    if out is None:
        ldc = (n_out + 7) // 8 * 8               # rows stay 16-byte aligned (e.g. the 32002-wide LM head)
        out = torch.empty((M, ldc), dtype=out_dtype, device=a.device)
        ret = out if ldc == n_out else out[:, :n_out]
    elif ldc is None:
        ldc = n_out
    ep = GemmEpilogue()
    ep.bias_bf16 = bias.data_ptr() if bias is not None else None
    ep.row_gate = row_gate.data_ptr() if row_gate is not None else None
    if row_gate is not None:
        assert row_gate.dtype == torch.float32 and row_gate.numel() >= M
    ep.residual = residual.data_ptr() if residual is not None else None
    ep.residual_dtype = _dt(residual) if residual is not None else 0
    ep.ld_res = (n_out if ld_res is None else ld_res)
    ep.act = ACT[act]
    ep.swiglu = 1 if swiglu else 0
    ep.use_scale = 0 if scale is None else 1
    ep.scale = 0.0 if scale is None else float(scale)
    ep.out_dtype = _dt(out)
    splits, ws_bytes = _splitk_plan(M, N, K) if SPLITK else (1, 0)
    if splits > 1:                      # skinny GEMM (student pass, decode steps): deterministic split-K through a workspace
        ws = _workspace(a.device, ws_bytes)
        _timed("gemm", 2.0 * M * N * K, lambda: check(_lib.lib().licv_gemm_bf16_splitk(
            _p(a), lda, _p(w), K, _p(out), ldc, M, N, K, C.byref(ep), splits, _p(ws), ws.numel(), _stream(a))), label=(M, N, K))
    else:
        _timed("gemm", 2.0 * M * N * K, lambda: check(_lib.lib().licv_gemm_bf16(
            _p(a), lda, _p(w), K, _p(out), ldc, M, N, K, C.byref(ep), _stream(a))), label=(M, N, K))
    return out if ret is None else ret


class Slices:
    """The fp32 split-K slices of a plain projection, left in the workspace for the row kernel behind it (licv_gemm_bf16_splitk_produce)."""
    def __init__(self, ws, splits, slice_elems, row_stride, rows, cols):
        self.ws, self.splits, self.slice_elems, self.row_stride, self.rows, self.cols = ws, splits, slice_elems, row_stride, rows, cols

    def args(self):
        return _p(self.ws), self.splits, self.slice_elems, self.row_stride


def linear_produce(a: torch.Tensor, w: torch.Tensor) -> Optional[Slices]:
    """The producer half of the split-K form of a @ w.T (None when the plan does not split the shape): the consumer — add_rmsnorm_ws_,
    inject_renorm_ws, rotary_kv_append — sums the slices itself, bit for bit what linear() would have written as bf16."""
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.is_contiguous() and a.is_contiguous()
    N, K = w.shape
    M = a.numel() // K
    splits, ws_bytes = _splitk_plan(M, N, K) if SPLITK else (1, 0)
    if splits <= 1:
        return None
    ws = _workspace(a.device, ws_bytes)
    se, rs = C.c_int64(0), C.c_int64(0)
    _timed("gemm", 2.0 * M * N * K, lambda: check(_lib.lib().licv_gemm_bf16_splitk_produce(
        _p(a), K, _p(w), K, M, N, K, splits, _p(ws), ws.numel(), C.byref(se), C.byref(rs), _stream(a))), label=(M, N, K))
    return Slices(ws, splits, se.value, rs.value, M, N)


def add_rmsnorm_ws_(h: torch.Tensor, branch: Slices, w: torch.Tensor, eps: float, flavour: int = 0,
                    row_gate: Optional[torch.Tensor] = None, scale: Optional[float] = None) -> torch.Tensor:
    """add_rmsnorm_ with the branch still in split-K slices."""
    dim = h.shape[-1]
    rows = h.numel() // dim
    assert h.is_contiguous() and branch.rows == rows and branch.cols == dim
    _bf16c(w, "w")
    out = torch.empty(h.shape, dtype=torch.bfloat16, device=h.device)
    check(_lib.lib().licv_add_rmsnorm_fwd_ws(_p(h), _dt(h), *branch.args(), _p(row_gate), 0 if scale is None else 1,
                                             0.0 if scale is None else float(scale), _p(w), _p(out), rows, dim, float(eps), flavour, _stream(h)))
    return out


def inject_renorm_ws(h: torch.Tensor, pre: Slices, icv_row: torch.Tensor, alpha: Optional[torch.Tensor], norm_weight: torch.Tensor,
                     norm_eps: float = 1e-6, out: Optional[torch.Tensor] = None):
    """inject_renorm(h, ..., pre=branch) with the branch still in split-K slices; returns (fp32 h', bf16 RMSNorm(h'))."""
    H = h.shape[-1]
    rows = h.numel() // H
    assert h.is_contiguous() and icv_row.dtype == torch.float32 and icv_row.numel() == H and pre.rows == rows and pre.cols == H
    _bf16c(norm_weight, "norm_weight")
    if out is None:
        out = torch.empty(h.shape, dtype=torch.float32, device=h.device)
    xn = torch.empty(h.shape, dtype=torch.bfloat16, device=h.device)
    check(_lib.lib().licv_inject_renorm_pre_fwd_ws(_p(h), _dt(h), *pre.args(), _p(icv_row), _p(alpha), _p(out), rows, H,
                                                   _p(norm_weight), _p(xn), float(norm_eps), _stream(h)))
    return out, xn


def rotary_kv_append(qkv, cos: torch.Tensor, sin: torch.Tensor, position_ids: torch.Tensor, batch: int, S: int, n_heads: int,
                     head_dim: int, cache: torch.Tensor, past: int, q_out: Optional[torch.Tensor] = None):
    """Rotary on the Q | K heads of a fused QKV projection and the append of K | V to cache (batch, max_len, 2H) at `past`.
    qkv: the (batch * S, 3H) bf16 rows (Q rotated in place), or the projection's split-K Slices (rotated Q written to q_out[:, :H])."""
    H = n_heads * head_dim
    assert cache.dtype == torch.bfloat16 and cache.is_contiguous() and cache.shape[0] == batch and cache.shape[2] == 2 * H
    if isinstance(qkv, Slices):
        assert q_out is not None and q_out.dtype == torch.bfloat16 and q_out.is_contiguous() and q_out.numel() == batch * S * 3 * H
        check(_lib.lib().licv_rotary_kv_append_ws(*qkv.args(), _p(q_out), _p(cos), _p(sin), _p(position_ids), batch, S, n_heads, head_dim,
                                                  cos.shape[0], _p(cache), cache.shape[1], past, _stream(cache)))
        return q_out
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.numel() == batch * S * 3 * H
    check(_lib.lib().licv_rotary_kv_append(_p(qkv), _p(cos), _p(sin), _p(position_ids), batch, S, n_heads, head_dim,
                                           cos.shape[0], _p(cache), cache.shape[1], past, _stream(cache)))
    return qkv


def decode_attn(qkv, cos: torch.Tensor, sin: torch.Tensor, position_ids: torch.Tensor, cache: torch.Tensor, past: int, n_heads: int,
                n_kv_heads: int, head_dim: int, scale: float, key_valid: Optional[torch.Tensor] = None,
                kv_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One decode step's attention block in one launch (licv_decode_attn): qkv = the (M, [Q | K | V]) bf16 rows of the fused projection or
    its split-K Slices; rotary, K | V appended to cache[r, past], attention over positions 0..past (read through kv_rows when given).
    cache: (rows >= M, max_len, 2 * n_kv_heads * head_dim) bf16.  Returns O (M, n_heads * head_dim) bf16."""
    from ._lib import DecodeAttnArgs
    kd = n_kv_heads * head_dim
    assert cache.dtype == torch.bfloat16 and cache.is_contiguous() and cache.shape[2] == 2 * kd
    a = DecodeAttnArgs()
    if isinstance(qkv, Slices):
        M = qkv.rows
        assert qkv.cols == (n_heads + 2 * n_kv_heads) * head_dim
        a.qkv_ws, a.splits, a.slice_elems, a.row_stride = qkv.ws.data_ptr(), qkv.splits, qkv.slice_elems, qkv.row_stride
    else:
        assert qkv.dtype == torch.bfloat16 and qkv.dim() == 2 and qkv.stride(1) == 1
        M = qkv.shape[0]
        a.qkv_bf16, a.ldq = qkv.data_ptr(), qkv.stride(0)
    assert cache.shape[0] >= M and position_ids.numel() == M and position_ids.dtype == torch.int64
    out = torch.empty((M, n_heads * head_dim), dtype=torch.bfloat16, device=cache.device)
    a.cos, a.sin, a.position_ids, a.n_pos = cos.data_ptr(), sin.data_ptr(), position_ids.data_ptr(), cos.shape[0]
    a.cache, a.max_len, a.past = cache.data_ptr(), cache.shape[1], int(past)
    if kv_rows is not None:
        assert kv_rows.dtype == torch.int32 and kv_rows.is_contiguous() and kv_rows.shape[0] >= M
        a.kv_rows, a.ld_kv_rows = kv_rows.data_ptr(), kv_rows.shape[1]
    if key_valid is not None:
        assert key_valid.dtype == torch.int32 and key_valid.is_contiguous() and key_valid.shape == (M, past + 1)
        a.key_valid = key_valid.data_ptr()
    a.out, a.M, a.n_heads, a.n_kv_heads, a.head_dim, a.scale = out.data_ptr(), M, n_heads, n_kv_heads, head_dim, float(scale)
    check(_lib.lib().licv_decode_attn(C.byref(a), _stream(cache)))
    return out


def quantize_fp8(x: torch.Tensor):
    """Per-row dynamic quantisation to OCP e4m3: returns (q uint8 (rows, K), scale fp32 (rows,))."""
    assert x.dim() == 2 and x.stride(1) == 1
    rows, K = x.shape
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(_lib.lib().licv_quantize_rows_fp8(_p(x), _dt(x), _p(q), _p(sc), rows, K, x.stride(0), K, _stream(x)))
    return q, sc


def _q8_bufs(rows: int, dim: int, device):
    return torch.empty((rows, dim), dtype=torch.uint8, device=device), torch.empty((rows,), dtype=torch.float32, device=device)


def rmsnorm_q8(x: torch.Tensor, w: torch.Tensor, eps: float, flavour: int = 0, want_bf16: bool = False):
    """RMSNorm whose rows leave as fp8: (q uint8 (rows, dim), scale fp32 (rows,)) == quantize_fp8(rmsnorm(x)); the bf16 rows
    themselves are written only if want_bf16 (third return value)."""
    dim = x.shape[-1]
    rows = x.numel() // dim
    assert x.is_contiguous()
    q, sc = _q8_bufs(rows, dim, x.device)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    check(_lib.lib().licv_rmsnorm_fwd_q8(_p(x), _dt(x), _p(w), _p(out), _p(q), _p(sc), rows, dim, float(eps), flavour, _stream(x)))
    return (q, sc, out) if want_bf16 else (q, sc)


def add_rmsnorm_q8_(h: torch.Tensor, branch: torch.Tensor, w: torch.Tensor, eps: float, flavour: int = 0):
    """h += branch (in place), returns the fp8 image (q, scale) of the RMSNorm of the new h (== quantize_fp8(add_rmsnorm_(h, branch, ...)))."""
    dim = h.shape[-1]
    rows = h.numel() // dim
    assert h.is_contiguous() and branch.is_contiguous() and branch.dtype == torch.bfloat16 and branch.numel() == h.numel()
    q, sc = _q8_bufs(rows, dim, h.device)
    check(_lib.lib().licv_add_rmsnorm_fwd_q8(_p(h), _dt(h), _p(branch), _p(w), None, _p(q), _p(sc), rows, dim, float(eps), flavour, _stream(h)))
    return q, sc


def inject_renorm_add_q8(branch: torch.Tensor, icv_row: torch.Tensor, residual: torch.Tensor, alpha: Optional[torch.Tensor],
                         norm_weight: torch.Tensor, norm_eps: float = 1e-6, norm_flavour: int = 1):
    """inject_renorm_add with the fused RMSNorm's rows leaving as fp8: returns (out fp32, q, scale)."""
    H = branch.shape[-1]
    assert branch.is_contiguous() and residual.is_contiguous() and residual.shape == branch.shape
    rows = branch.numel() // H
    out = torch.empty(branch.shape, dtype=torch.float32, device=branch.device)
    q, sc = _q8_bufs(rows, H, branch.device)
    check(_lib.lib().licv_inject_renorm_add_fwd_q8(_p(branch), _dt(branch), _p(icv_row), _p(alpha), _p(residual), _dt(residual), _p(out), rows, H,
                                                   _p(norm_weight), None, _p(q), _p(sc), float(norm_eps), norm_flavour, _stream(branch)))
    return out, q, sc


def layernorm_q8(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float):
    """LayerNorm whose rows leave as fp8 only: (q, scale) == quantize_fp8(layernorm(x))."""
    dim = x.shape[-1]
    rows = x.numel() // dim
    assert x.is_contiguous() and x.dtype == torch.bfloat16
    q, sc = _q8_bufs(rows, dim, x.device)
    check(_lib.lib().licv_layernorm_fwd_q8(_p(x), _p(w), _p(b), None, _p(q), _p(sc), rows, dim, float(eps), _stream(x)))
    return q, sc


def linear_fp8(aq: torch.Tensor, a_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None,
               act=None, swiglu: bool = False, row_gate: Optional[torch.Tensor] = None, scale: Optional[float] = None,
               residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None):
    """out = epilogue((aq @ wq.T) * a_scale[:, None] * w_scale[None, :]) on the fp8 MFMA; aq (M, K), wq (N, K) uint8 e4m3."""
    assert aq.dtype == torch.uint8 and wq.dtype == torch.uint8 and aq.is_contiguous() and wq.is_contiguous()
    M, K = aq.shape
    N = wq.shape[0]
    n_out = N // 2 if swiglu else N
    if out_dtype is None:
        out_dtype = residual.dtype if residual is not None else torch.bfloat16
    ret = None
    if out is None:
        ldc = (n_out + 7) // 8 * 8
        out = torch.empty((M, ldc), dtype=out_dtype, device=aq.device)
        ret = out if ldc == n_out else out[:, :n_out]
    else:
        ldc = out.stride(0)
    ep = GemmEpilogue()
    ep.bias_bf16 = bias.data_ptr() if bias is not None else None
    ep.row_gate = row_gate.data_ptr() if row_gate is not None else None
    ep.residual = residual.data_ptr() if residual is not None else None
    ep.residual_dtype = _dt(residual) if residual is not None else 0
    ep.ld_res = residual.stride(0) if residual is not None else n_out
    ep.act = ACT[act]
    ep.swiglu = 1 if swiglu else 0
    ep.use_scale = 0 if scale is None else 1
    ep.scale = 0.0 if scale is None else float(scale)
    ep.out_dtype = _dt(out)
    _timed("gemm", 2.0 * M * N * K, lambda: check(_lib.lib().licv_gemm_fp8(
        _p(aq), K, _p(a_scale), _p(wq), K, _p(w_scale), _p(out), ldc, M, N, K, C.byref(ep), _stream(aq))), label=(M, N, K))
    return out if ret is None else ret


def pack_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    _bf16c(gate, "gate"); _bf16c(up, "up")
    inter, K = gate.shape
    out = torch.empty((2 * inter, K), dtype=torch.bfloat16, device=gate.device)
    check(_lib.lib().licv_pack_gate_up(_p(gate), _p(up), _p(out), inter, K, _stream(gate)))
    return out


# ------------------------------------------------------------------------------------------ attention
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, Sq: int, Sk: int, n_heads: int, n_kv_heads: int,
              head_dim: int, q_bs: int, q_rs: int, kv_bs: int, kv_rs: int, scale: float, mask_mode: int = 0,
              key_valid: Optional[torch.Tensor] = None, img_mask: Optional[torch.Tensor] = None, img_len: int = 0,
              out: Optional[torch.Tensor] = None):
    """q/k/v are (possibly offset views into) bf16 buffers; strides are in elements.  Returns (B, Sq, nh*hd) bf16."""
    if out is None:
        out = torch.empty((B, Sq, n_heads * head_dim), dtype=torch.bfloat16, device=q.device)
    a = AttnArgs()
    a.q, a.q_bs, a.q_rs = q.data_ptr(), q_bs, q_rs
    a.k, a.v, a.kv_bs, a.kv_rs = k.data_ptr(), v.data_ptr(), kv_bs, kv_rs
    a.o = out.data_ptr()
    a.B, a.Sq, a.Sk, a.n_heads, a.n_kv_heads, a.head_dim = B, Sq, Sk, n_heads, n_kv_heads, head_dim
    a.scale, a.mask_mode = float(scale), mask_mode
    if key_valid is not None:
        assert key_valid.dtype == torch.int32 and key_valid.is_contiguous()
    a.key_valid = key_valid.data_ptr() if key_valid is not None else None
    if img_mask is not None:
        assert img_mask.dtype == torch.int32 and img_mask.is_contiguous()
        a.img_mask, a.n_img, a.img_len = img_mask.data_ptr(), img_mask.shape[-1], img_len
    else:
        a.img_mask, a.n_img, a.img_len = None, 0, 0
    check(_lib.lib().licv_attn_fwd(C.byref(a), _stream(q)))
    return out


# ------------------------------------------------------------------------------------------ gathers
def embed_gather(ids: torch.Tensor, table: torch.Tensor, extra: Optional[torch.Tensor], vocab: int) -> torch.Tensor:
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    dim = table.shape[1]
    out = torch.empty((*ids.shape, dim), dtype=torch.bfloat16, device=table.device)
    check(_lib.lib().licv_embed_gather(_p(ids), _p(table), _p(extra), _p(out), ids.numel(), dim, vocab,
                                       0 if extra is None else extra.shape[0], _stream(table)))
    return out


def im2col_patches(pix: torch.Tensor, patch: int, ld_out: int) -> torch.Tensor:
    """pix: (n_img, 3, H, W) bf16 contiguous -> (n_img*gh*gw, ld_out) bf16, zero padded past 3*patch^2."""
    _bf16c(pix, "pixel_values")
    n, c, H, W = pix.shape
    assert c == 3
    out = torch.empty((n * (H // patch) * (W // patch), ld_out), dtype=torch.bfloat16, device=pix.device)
    check(_lib.lib().licv_im2col_patches(_p(pix), _p(out), n, H, W, patch, ld_out, _stream(pix)))
    return out


def vit_embed_ln(patches: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, w: torch.Tensor, b: torch.Tensor,
                 n_img: int, n_patch: int, eps: float) -> torch.Tensor:
    dim = cls.numel()
    out = torch.empty((n_img, n_patch + 1, dim), dtype=torch.bfloat16, device=patches.device)
    check(_lib.lib().licv_vit_embed_ln(_p(patches), _p(cls), _p(pos), _p(w), _p(b), _p(out), n_img, n_patch, dim,
                                       float(eps), _stream(patches)))
    return out


def tile_rows(src: torch.Tensor, rows: int) -> torch.Tensor:
    _bf16c(src, "src")
    period, dim = src.shape
    out = torch.empty((rows, dim), dtype=torch.bfloat16, device=src.device)
    check(_lib.lib().licv_tile_rows(_p(src), _p(out), rows, dim, period, _stream(src)))
    return out


def swiglu(gu: torch.Tensor) -> torch.Tensor:
    _bf16c(gu, "gu")
    rows, two_i = gu.shape
    out = torch.empty((rows, two_i // 2), dtype=torch.bfloat16, device=gu.device)
    check(_lib.lib().licv_swiglu(_p(gu), _p(out), rows, two_i // 2, _stream(gu)))
    return out


# ------------------------------------------------------------------------------------------ loss / optim
def kl_rows(stu: torch.Tensor, tea: torch.Tensor, stu_rows: torch.Tensor, tea_rows: torch.Tensor, vocab: int,
            temperature: float, eps: float) -> torch.Tensor:
    """stu/tea: 2-D logits buffers (rows x ld); *_rows: int64 row indices (same count).  Returns fp32 per-row KL."""
    assert stu.dtype == tea.dtype and stu.dim() == 2 and tea.dim() == 2
    n = stu_rows.numel()
    out = torch.empty((n,), dtype=torch.float32, device=stu.device)
    check(_lib.lib().licv_kl_rows_fwd(_p(stu), _p(tea), _dt(stu), _p(stu_rows), _p(tea_rows), n, vocab, stu.stride(0),
                                      tea.stride(0), float(temperature), float(eps), _p(out), _stream(stu)))
    return out


def kl_rows_dtemp(stu: torch.Tensor, tea: torch.Tensor, stu_rows: torch.Tensor, tea_rows: torch.Tensor, vocab: int,
                  temperature: float, eps: float) -> torch.Tensor:
    """d/dT of kl_rows' per-row sums (fp32), for a trainable temperature (ref:icv_src/icv_module.py:49-52)."""
    assert stu.dtype == tea.dtype and stu.dim() == 2 and tea.dim() == 2
    n = stu_rows.numel()
    out = torch.empty((n,), dtype=torch.float32, device=stu.device)
    check(_lib.lib().licv_kl_rows_dtemp(_p(stu), _p(tea), _dt(stu), _p(stu_rows), _p(tea_rows), n, vocab, stu.stride(0),
                                        tea.stride(0), float(temperature), float(eps), _p(out), _stream(stu)))
    return out


def adamw_step_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, n_group0: int, lr0: float, lr1: float,
                step: int, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-3, grad_scale=1.0):
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous()
    check(_lib.lib().licv_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), n_group0, lr0, lr1, beta1, beta2, eps,
                                     weight_decay, step, grad_scale, _stream(p)))
    return p


# ------------------------------------------------------------------------------------------ backward (student pass)
def rmsnorm_bwd(x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor, eps: float, accumulate: bool,
                inner: int = 1, ld_x: Optional[int] = None, ld_dy: Optional[int] = None, ld_dx: Optional[int] = None,
                rows: Optional[int] = None, dim: Optional[int] = None, flavour: int = 0):
    """flavour as in rmsnorm(): 0 Idefics, 1 Mistral (decides whether d(w * xhat) is a bf16 or an fp32 product, see licv_hip.h)."""
    dim = x.shape[-1] if dim is None else dim
    rows = x.numel() // dim if rows is None else rows
    ld_x = dim * inner if ld_x is None else ld_x
    ld_dy = dim * inner if ld_dy is None else ld_dy
    ld_dx = dim * inner if ld_dx is None else ld_dx
    check(_lib.lib().licv_rmsnorm_bwd(_p(x), _dt(x), _p(w), _p(dy), _dt(dy), _p(dx), _dt(dx), rows, dim, inner, ld_x, ld_dy, ld_dx,
                                      float(eps), 1 if accumulate else 0, flavour, _stream(x)))
    return dx


def rmsnorm_bwd_from(x: torch.Tensor, w: torch.Tensor, a: torch.Tensor, wt: torch.Tensor, dx: torch.Tensor, eps: float, accumulate: bool,
                     flavour: int = 0):
    """rmsnorm_bwd(x, w, dy = a @ wt.T, dx, ...) with the dgrad GEMM and the norm's backward fused at the launch level: where the plan
    splits K (the 256-row student always does) only the producer half of the GEMM runs and the norm kernel sums the slices itself -
    bit for bit rmsnorm_bwd(x, w, linear(a, wt), ...), one launch and one bf16 round trip of dy less."""
    dim = x.shape[-1]
    sl = linear_produce(a, wt) if (dim >= 1024 and x.is_contiguous() and dx.is_contiguous()) else None
    if sl is None:
        return rmsnorm_bwd(x, w, linear(a, wt), dx, eps, accumulate, flavour=flavour)
    rows = x.numel() // dim
    assert sl.rows == rows and sl.cols == dim
    check(_lib.lib().licv_rmsnorm_bwd_ws(_p(x), _dt(x), _p(w), *sl.args(), _p(dx), _dt(dx), rows, dim, float(eps), 1 if accumulate else 0,
                                         flavour, _stream(x)))
    return dx


def swiglu_bwd(gu: torch.Tensor, dact: torch.Tensor) -> torch.Tensor:
    rows, two_i = gu.shape
    out = torch.empty_like(gu)
    check(_lib.lib().licv_swiglu_bwd(_p(gu), _p(dact), _p(out), rows, two_i // 2, _stream(gu)))
    return out


def branch_grad(dh: torch.Tensor, scale: Optional[float] = None, row_gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert dh.dtype == torch.float32 and dh.is_contiguous()
    rows, dim = dh.shape
    out = torch.empty((rows, dim), dtype=torch.bfloat16, device=dh.device)
    check(_lib.lib().licv_branch_grad(_p(dh), _p(out), rows, dim, 0.0 if scale is None else float(scale), 0 if scale is None else 1,
                                      _p(row_gate), _stream(dh)))
    return out


def attention_bwd_small(q, k, v, dout, B, Sq, Sk, n_heads, n_kv_heads, head_dim, q_bs, q_rs, kv_bs, kv_rs, scale, mask_mode,
                        dq, dq_bs, dq_rs, dk=None, dv=None, dkv_bs=0, dkv_rs=0, key_valid=None, img_mask=None, img_len=0):
    a = AttnArgs()
    a.q, a.q_bs, a.q_rs = q.data_ptr(), q_bs, q_rs
    a.k, a.v, a.kv_bs, a.kv_rs = k.data_ptr(), v.data_ptr(), kv_bs, kv_rs
    a.o = None
    a.B, a.Sq, a.Sk, a.n_heads, a.n_kv_heads, a.head_dim = B, Sq, Sk, n_heads, n_kv_heads, head_dim
    a.scale, a.mask_mode = float(scale), mask_mode
    a.key_valid = key_valid.data_ptr() if key_valid is not None else None
    if img_mask is not None:
        a.img_mask, a.n_img, a.img_len = img_mask.data_ptr(), img_mask.shape[-1], img_len
    else:
        a.img_mask, a.n_img, a.img_len = None, 0, 0
    assert dout.is_contiguous() and dout.dtype == torch.bfloat16
    check(_lib.lib().licv_attn_bwd_small(C.byref(a), _p(dout), _p(dq), dq_bs, dq_rs, _p(dk), _p(dv), dkv_bs, dkv_rs, _stream(q)))
    return dq


def ce_rows(logits: torch.Tensor, rows: torch.Tensor, labels: torch.Tensor, vocab: int, grad: Optional[torch.Tensor] = None,
            grad_coef: float = 0.0, grad_rows: Optional[torch.Tensor] = None, accumulate: bool = False, want_loss: bool = True,
            grad_coef_dev: Optional[torch.Tensor] = None):
    """Cross-entropy of logits[rows] against labels (fp32 maths).  Returns per-row losses (fp32) or None; optionally
    writes / accumulates grad_coef * (softmax - onehot) into grad[(grad_rows or arange)] (bf16, row stride grad.stride(0));
    grad_coef_dev: optional 1-element fp32 device tensor multiplied into grad_coef inside the kernel (no host sync)."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and rows.dtype == torch.int64 and labels.dtype == torch.int64
    n = rows.numel()
    out = torch.empty((n,), dtype=torch.float32, device=logits.device) if want_loss else None
    check(_lib.lib().licv_ce_rows(_p(logits), _dt(logits), _p(rows.contiguous()), _p(labels.contiguous()), n, vocab, logits.stride(0), _p(out),
                                  float(grad_coef), _p(grad_coef_dev), _p(grad), grad.stride(0) if grad is not None else 0,
                                  _p(grad_rows.contiguous()) if grad_rows is not None else None, 1 if accumulate else 0, _stream(logits)))
    return out


def head_group_sum(src: torch.Tensor, out: torch.Tensor, rows: int, n_groups: int, rep: int, head_dim: int, ld_src: int, ld_out: int):
    """Backward of repeat_kv: sums each group of `rep` query heads (bf16 in, fp32 accumulate, bf16 out)."""
    check(_lib.lib().licv_head_group_sum(_p(src), _p(out), rows, n_groups, rep, head_dim, ld_src, ld_out, _stream(src)))
    return out


def kl_rows_bwd(stu, tea, stu_rows, tea_rows, vocab, temperature, eps, upstream=1.0, upstream_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Returns (n_rows, vocab_padded_to_8) bf16 with zero pad columns (so it can feed the head's dgrad GEMM directly).
    upstream_dev: optional 1-element fp32 device tensor multiplied into the coefficient inside the kernel."""
    n = stu_rows.numel()
    ld = (vocab + 7) // 8 * 8
    grad = torch.zeros((n, ld), dtype=torch.bfloat16, device=stu.device)
    check(_lib.lib().licv_kl_rows_bwd(_p(stu), _p(tea), _dt(stu), _p(stu_rows), _p(tea_rows), n, vocab, stu.stride(0), tea.stride(0),
                                      float(temperature), float(eps), float(upstream), _p(upstream_dev), _p(grad), ld, _stream(stu)))
    return grad
