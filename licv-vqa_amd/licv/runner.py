"""ctypes side of the native layer runner (csrc/runner.hip, ``licv_idefics_text_forward``): the Idefics language stack of
``IdeficsEngine.forward`` — same kernels, same order, same results — issued from one C call instead of ~12 Python-level
launches per layer.  Used by the engine whenever nothing has to be captured (inference, generate, the teacher pass)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import check

P, I64, F = C.c_void_p, C.c_int64, C.c_float


class _DecW(C.Structure):
    _fields_ = [(n, P) for n in ("in_ln", "qkv_w", "o_w", "post_ln", "gu_w", "down_w")]


class _XatW(C.Structure):
    _fields_ = [(n, P) for n in ("in_ln", "q_w", "kv_w", "o_w", "qn_w", "kn_w", "post_ln", "gu_w", "down_w")] + [("gate_attn", F), ("gate_dense", F)]


class _Weights(C.Structure):
    _fields_ = [(n, I64) for n in ("hidden", "inter", "n_heads", "head_dim", "n_layers", "cross_interval", "vocab", "n_extra_vocab",
                                   "vocab_total", "img_dim", "img_len", "rope_len")] + [("rms_eps", F)] + \
               [(n, P) for n in ("embed", "embed_extra", "final_ln", "lm_head", "cos", "sin")] + [("dec", C.POINTER(_DecW)), ("xat", C.POINTER(_XatW))]


class _Call(C.Structure):
    _fields_ = [(n, P) for n in ("input_ids", "key_valid", "position_ids", "image_states", "img_mask", "gate")] + \
               [(n, I64) for n in ("B", "S", "Sk", "Nk", "n_img")] + [("icv", P), ("alpha", P), ("hook_slot", C.POINTER(C.c_int32))] + \
               [("kv_cache", C.POINTER(P)), ("cache_max_len", I64), ("past", I64), ("kv_rows", P), ("ld_kv_rows", I64),
                ("xkv_cached", C.POINTER(P)), ("xkv_out", C.POINTER(P)),
                ("logits_rows", P), ("n_rows", I64)] + \
               [(n, P) for n in ("h16", "h32", "x", "xn", "q", "qkv", "o", "act", "xkv", "xsel")] + \
               [("workspace", P), ("workspace_bytes", I64), ("logits", P), ("ld_logits", I64)]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class TextRunner:
    def __init__(self, weights):
        a, w = weights.arch, weights
        self.arch, self.w = a, w
        n_x = len(w.xat)
        self._dec = (_DecW * a.num_layers)(*[_DecW(*(t.data_ptr() for t in (D.in_ln, D.qkv_w, D.o_w, D.post_ln, D.gu_w, D.down_w))) for D in w.dec])
        self._xat = (_XatW * max(n_x, 1))(*[_XatW(X.in_ln.data_ptr(), X.q_w.data_ptr(), X.kv_w.data_ptr(), X.o_w.data_ptr(), _ptr(X.qn_w),
                                                   _ptr(X.kn_w), X.post_ln.data_ptr(), X.gu_w.data_ptr(), X.down_w.data_ptr(),
                                                   float(X.gate_attn), float(X.gate_dense)) for X in w.xat])
        W = _Weights()
        W.hidden, W.inter, W.n_heads, W.head_dim = a.hidden_size, a.intermediate_size, a.num_heads, a.head_dim
        W.n_layers, W.cross_interval = a.num_layers, a.cross_layer_interval
        W.vocab, W.n_extra_vocab, W.vocab_total = a.vocab_size, (0 if w.embed_extra is None else w.embed_extra.shape[0]), w.lm_head.shape[0]
        W.img_dim, W.img_len, W.rope_len, W.rms_eps = a.v_embed, a.image_seq_len, w.cos.shape[0], a.rms_eps
        W.embed, W.embed_extra, W.final_ln, W.lm_head, W.cos, W.sin = (w.embed.data_ptr(), _ptr(w.embed_extra), w.final_ln.data_ptr(),
                                                                      w.lm_head.data_ptr(), w.cos.data_ptr(), w.sin.data_ptr())
        W.dec, W.xat = self._dec, self._xat
        self._W = W
        lib = _lib.lib()
        lib.licv_idefics_text_forward.argtypes = [C.POINTER(_Weights), C.POINTER(_Call), P]
        lib.licv_idefics_text_forward.restype = C.c_int
        lib.licv_workspace_size.restype = C.c_int64
        self._fn = lib.licv_idefics_text_forward
        self._ws = {}

    def _workspace(self, M: int, rows: int, dev, xkv_rows: int = 0):
        a = self.arch
        H, I, V = a.hidden_size, a.intermediate_size, self.w.lm_head.shape[0]
        lib = _lib.lib()
        shapes = [(M, H, H), (M, 3 * H, H), (M, 2 * I, H), (M, H, I), (rows, V, H)]
        if xkv_rows > 0:                                    # the cross-attention K|V projection on the image rows (B * Nk, 2H, img_dim)
            shapes.append((xkv_rows, 2 * H, a.v_embed))
        need = max(int(lib.licv_workspace_size(m, n, k)) for (m, n, k) in shapes)
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        buf = self._ws.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty((max(need, 1 << 20),), dtype=torch.uint8, device=dev)
            self._ws[key] = buf
        return buf

    def forward(self, input_ids, key_valid, position_ids, image_states, img_mask, gate, icv=None, alpha=None,
                hook_layers: Optional[Sequence[int]] = None, kv_cache=None, logits_rows=None):
        """All tensors on the device, contiguous, in the dtypes of IdeficsEngine.forward's locals.  Returns the logits exactly as
        that method does: (B, S, V) view (or (n_rows, V) with logits_rows)."""
        a, w = self.arch, self.w
        dev = w.device
        B, S = input_ids.shape
        M, H, I = B * S, a.hidden_size, a.intermediate_size
        Nk = image_states.shape[1]
        V = w.lm_head.shape[0]
        ldc = (V + 7) // 8 * 8
        rows = M if logits_rows is None else int(logits_rows.numel())
        bf = dict(dtype=torch.bfloat16, device=dev)
        keep = []                                                       # scratch tensors must outlive the (asynchronous) launches: the
        new = lambda *shape, **kw: (keep.append(torch.empty(shape, **kw)), keep[-1])[1]      # stream-ordered allocator guarantees that

        c = _Call()
        c.input_ids, c.key_valid, c.position_ids = input_ids.data_ptr(), key_valid.data_ptr(), position_ids.data_ptr()
        c.image_states, c.img_mask, c.gate = image_states.data_ptr(), img_mask.data_ptr(), gate.data_ptr()
        c.B, c.S, c.Sk, c.Nk, c.n_img = B, S, key_valid.shape[1], Nk, img_mask.shape[-1]
        slots = None
        if icv is not None and hook_layers is not None:
            c.icv = icv.data_ptr()
            c.alpha = _ptr(alpha)
            slots = (C.c_int32 * a.num_layers)(*([-1] * a.num_layers))
            for i, l in enumerate(hook_layers):
                slots[int(l)] = i
            c.hook_slot = slots
        n_x = len(w.xat)
        kv_arr = xc_arr = xo_arr = None
        if kv_cache is not None:
            kv_arr = (P * a.num_layers)(*[t.data_ptr() for t in kv_cache.kv])
            c.kv_cache, c.cache_max_len, c.past = kv_arr, kv_cache.max_len, kv_cache.len
            if S == 1 and getattr(kv_cache, "rows", None) is not None:   # decode step of a beam search: history read through the row table
                c.kv_rows, c.ld_kv_rows = kv_cache.rows.data_ptr(), kv_cache.rows.shape[1]
            if getattr(kv_cache, "xkv", None) is not None:               # cross-attention K|V projected at the prefill
                xc_arr = (P * n_x)(*[t.data_ptr() for t in kv_cache.xkv])
                c.xkv_cached = xc_arr
            elif kv_cache.len == 0:
                kv_cache.xkv = [torch.empty((B, Nk, 2 * H), **bf) for _ in range(n_x)]
                xo_arr = (P * n_x)(*[t.data_ptr() for t in kv_cache.xkv])
                c.xkv_out = xo_arr
        c.h16, c.h32 = new(M, H, **bf).data_ptr(), new(M, H, dtype=torch.float32, device=dev).data_ptr()
        c.x, c.xn, c.q, c.o = (new(M, H, **bf).data_ptr() for _ in range(4))
        c.qkv, c.act = new(M, 3 * H, **bf).data_ptr(), new(M, I, **bf).data_ptr()
        if c.xkv_cached is None or not bool(c.xkv_cached):
            c.xkv = new(B * Nk, 2 * H, **bf).data_ptr()
        if logits_rows is not None:
            c.logits_rows, c.n_rows, c.xsel = logits_rows.data_ptr(), rows, new(rows, H, **bf).data_ptr()
        ws = self._workspace(M, rows, dev, B * Nk)
        c.workspace, c.workspace_bytes = ws.data_ptr(), ws.numel()
        logits = torch.empty((rows, ldc), **bf)
        c.logits, c.ld_logits = logits.data_ptr(), ldc
        check(self._fn(C.byref(self._W), C.byref(c), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if kv_cache is not None:
            kv_cache.len += S
        out = logits if ldc == V else logits[:, :V]
        if logits_rows is not None:
            return out
        return out.view(B, S, V) if out.is_contiguous() else out.as_strided((B, S, V), (S * ldc, ldc, 1))
