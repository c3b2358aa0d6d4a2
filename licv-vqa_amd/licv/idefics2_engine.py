"""Native Idefics2 forward for MI355X with the ICV hook on every text layer's MLP branch.

Replaces, for the hot path, ``Idefics2ForConditionalGeneration.forward`` (hf:idefics2/modeling_idefics2.py:817-1010;
text model hf:mistral/modeling_mistral.py) run under ``torch.autocast(bf16)`` as the reference does for this model
(ref:icv_src/icv_module.py:36-56), plus the baukit hook on ``model.model.text_model.layers.N.mlp``
(ref:config/lmm/idefics2-8B-base.yaml:8).  Arithmetic = HIP kernels behind the C-ABI; torch supplies memory, the stream
and the integer plumbing of the NaViT tower (which images are padding, patch validity, bucketised position ids).

Layout in HBM: tokens flattened to rows.  Vision tokens (n_real_images * T, 1152) with T = (H/14)*(W/14) of the padded
batch resolution and a per-image key-valid mask; connector context (n*T, 4096); latents (n*64, 4096); the K/V input of a
perceiver layer is ONE (n, T+64, 4096) buffer the two RMSNorms write into directly.  Text side: fused QKV projection
((32+8+8)*128 x 4096, GQA heads read through strides), gate|up interleaved for the SwiGLU epilogue; the residual stream
is bf16 until the first hooked layer and fp32 after it (the hook's fp32 output promotes it, exactly as under autocast).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import frontend, ops
from .config import Idefics2Arch
from .idefics_engine import KVCache, _bf, _pad_cols


@dataclass
class _SigLayer:
    ln1_w: torch.Tensor; ln1_b: torch.Tensor; qkv_w: torch.Tensor; qkv_b: torch.Tensor
    out_w: torch.Tensor; out_b: torch.Tensor; ln2_w: torch.Tensor; ln2_b: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor; fc2_w: torch.Tensor; fc2_b: torch.Tensor
    q8: Optional[dict] = None           # fp8 copies of the four projection weights when fp8_vision


@dataclass
class _PercLayer:
    lat_ln: torch.Tensor; ctx_ln: torch.Tensor; q_w: torch.Tensor; kv_w: torch.Tensor; o_w: torch.Tensor
    post_ln: torch.Tensor; gu_w: torch.Tensor; down_w: torch.Tensor


@dataclass
class _TextLayer:
    in_ln: torch.Tensor; qkv_w: torch.Tensor; o_w: torch.Tensor; post_ln: torch.Tensor
    gu_w: torch.Tensor; down_w: torch.Tensor
    q8: Optional[dict] = None           # fp8 copies {name: (e4m3 bytes (N, K), per-output-channel scale (N,))} when fp8_text


class Idefics2Weights:
    """Engine-layout weights from an HF-named ``Idefics2ForConditionalGeneration`` state dict (bf16)."""

    def __init__(self, sd: Dict[str, torch.Tensor], arch: Idefics2Arch, device="cuda", max_positions: int = 4096,
                 fp8_text: bool = False, fp8_vision: bool = False):
        """fp8_text (BASELINE configs[4], "fp8 weights"): the text stack's projection weights are ALSO stored as OCP e4m3 with one
        scale per output channel; GEMMs with >= 512 rows then take their input as e4m3 rows with one scale per row (written by the
        norm kernel that produces them, or by the row quantiser) and run on the fp8 MFMA.  fp8_vision: the same for the four
        projections of every SigLIP layer."""
        a, dev = arch, torch.device(device)
        self.arch, self.device, self.fp8_text, self.fp8_vision = arch, dev, fp8_text, fp8_vision
        g = lambda k: _bf(sd[k], dev)
        cat = lambda p, names, suf: torch.cat([g(p + f"{n}.{suf}") for n in names]).contiguous()
        vp = "model.vision_model."
        self.patch_ld = (3 * a.v_patch * a.v_patch + 63) // 64 * 64
        self.patch_w = _pad_cols(g(vp + "embeddings.patch_embedding.weight").flatten(1), self.patch_ld)
        self.patch_b = g(vp + "embeddings.patch_embedding.bias")
        self.pos = g(vp + "embeddings.position_embedding.weight")
        self.vit: List[_SigLayer] = []
        # MLP width padded to a multiple of 64 (4304 -> 4352) with zero rows / bias / columns: gelu(0) = 0, so the result
        # is unchanged and fc2's K dimension qualifies for the 256x256 LDS-DMA GEMM (K % 64 == 0)
        ipad = (a.v_inter + 63) // 64 * 64

        def pad_rows(t):
            out = torch.zeros((ipad, *t.shape[1:]), dtype=t.dtype, device=t.device)
            out[: t.shape[0]] = t
            return out
        for i in range(a.v_layers):
            p = f"{vp}encoder.layers.{i}."
            qkv = ("q_proj", "k_proj", "v_proj")
            self.vit.append(_SigLayer(g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"),
                                      cat(p + "self_attn.", qkv, "weight"), cat(p + "self_attn.", qkv, "bias"),
                                      g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"),
                                      g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"),
                                      pad_rows(g(p + "mlp.fc1.weight")), pad_rows(g(p + "mlp.fc1.bias")),
                                      _pad_cols(g(p + "mlp.fc2.weight"), ipad), g(p + "mlp.fc2.bias")))
        if fp8_vision:
            for L in self.vit:
                L.q8 = {n: ops.quantize_fp8(getattr(L, n)) for n in ("qkv_w", "out_w", "fc1_w", "fc2_w")}
        self.post_ln_w, self.post_ln_b = g(vp + "post_layernorm.weight"), g(vp + "post_layernorm.bias")
        cp = "model.connector."
        mp = cp + "modality_projection."
        self.mp_gu = ops.pack_gate_up(g(mp + "gate_proj.weight"), g(mp + "up_proj.weight"))
        self.mp_down = g(mp + "down_proj.weight")
        rp = cp + "perceiver_resampler."
        self.latents = g(rp + "latents")
        self.perc: List[_PercLayer] = []
        for i in range(a.r_depth):
            p = f"{rp}layers.{i}."
            self.perc.append(_PercLayer(g(p + "input_latents_norm.weight"), g(p + "input_context_norm.weight"),
                                        g(p + "self_attn.q_proj.weight"), cat(p + "self_attn.", ("k_proj", "v_proj"), "weight"),
                                        g(p + "self_attn.o_proj.weight"), g(p + "post_attention_layernorm.weight"),
                                        ops.pack_gate_up(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")),
                                        g(p + "mlp.down_proj.weight")))
        self.perc_ln = g(rp + "norm.weight")
        tp = "model.text_model."
        self.embed = g(tp + "embed_tokens.weight")
        self.text: List[_TextLayer] = []
        for i in range(a.num_layers):
            p = f"{tp}layers.{i}."
            self.text.append(_TextLayer(g(p + "input_layernorm.weight"), cat(p + "self_attn.", ("q_proj", "k_proj", "v_proj"), "weight"),
                                        g(p + "self_attn.o_proj.weight"), g(p + "post_attention_layernorm.weight"),
                                        ops.pack_gate_up(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")),
                                        g(p + "mlp.down_proj.weight")))
        if fp8_text:
            ok = lambda w_: w_.shape[1] >= 256 and w_.shape[1] % 64 == 0
            for L in self.text:
                L.q8 = {n: ops.quantize_fp8(getattr(L, n)) for n in ("qkv_w", "o_w", "gu_w", "down_w") if ok(getattr(L, n))}
        self.final_ln = g(tp + "norm.weight")
        self.lm_head = g("lm_head.weight")
        # rotary tables: fp32 inv_freq and angles, cast to the model dtype (hf:mistral/modeling_mistral.py MistralRotaryEmbedding)
        hd = a.head_dim
        inv = 1.0 / (a.rope_base ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
        emb = torch.cat((torch.outer(torch.arange(max_positions, dtype=torch.float), inv),) * 2, dim=-1)
        self.cos, self.sin = _bf(emb.cos(), dev), _bf(emb.sin(), dev)
        self.max_positions = max_positions


class KVCache2(KVCache):
    """Per-layer (rows, max_len, [K | V] = 2 * n_kv_heads * head_dim) bf16 cache for hooked generate; see KVCache (row table, no moves)."""

    def __init__(self, arch: Idefics2Arch, batch: int, max_len: int, device, beams: int = 1):
        super().__init__(arch, batch, max_len, device, beams=beams, width=2 * arch.num_kv_heads * arch.head_dim)


class _HostFlags:
    """What the HOST must know about a batch before it can launch (how many images are real, whether any patch is masked, how
    many <image> tokens there are) is computed on the device (licv.frontend) and read back ONCE per distinct input: later
    forwards over the same tensor OBJECTS at the same version counter — the steady state of repeated evaluation, benchmarks
    and the teacher / student pair of a step — make no host round trip at all.  Entries hold weak references and are matched by
    object identity, so a new tensor that happens to reuse a freed address never hits a stale entry."""

    def __init__(self, keep: int = 16):
        self._entries, self._keep = [], keep

    @staticmethod
    def _sig(tensors):
        """Version counters of the tensors, or None when any of them has none to read: tensors made under
        `torch.inference_mode()` (ref:inference.py:246,300,324 wrap every entry point in it) do not track one, and a write
        to such a tensor could not be seen — those inputs are never cached (one 16-byte read-back per forward instead)."""
        sig = []
        for t in tensors:
            if t is None:
                sig.append(None)
                continue
            if t.is_inference():
                return None
            try:
                sig.append(t._version)
            except RuntimeError:
                return None
        return tuple(sig)

    def clear(self):
        """Drop every entry: for callers that rewrite an input through a path the version counter does not see
        (`.data`, `set_`, DLPack / numpy aliases, a custom kernel)."""
        self._entries.clear()

    def get(self, *tensors):
        sig = self._sig(tensors)
        if sig is None:
            return None
        for refs, vers, value in self._entries:
            if vers == sig and len(refs) == len(tensors) and all((r is None and t is None) or (r is not None and r() is t)
                                                                   for r, t in zip(refs, tensors)):
                return value
        return None

    def put(self, value, *tensors):
        import weakref
        if self._sig(tensors) is None:
            return value
        if len(self._entries) >= self._keep:
            self._entries.pop(0)
        self._entries.append((tuple(None if t is None else weakref.ref(t) for t in tensors), self._sig(tensors), value))
        return value


class Idefics2Engine:
    def __init__(self, weights: Idefics2Weights, fuse_hook_norm: bool = True):
        self.w, self.arch = weights, weights.arch
        self.fuse_hook_norm = fuse_hook_norm
        self._flags = _HostFlags()
        # batch slices on HIP streams of their own, as IdeficsEngine.batch_streams (same kernels, bit-identical logits; the partly
        # filled last round of one slice's GEMMs is filled by the other slice's workgroups).  0 / 1 = off.
        self.batch_streams = 2
        self._side_streams = []
        self._slice_views = {}
        self.fold_residual = True              # o-projection residual add folded into the post-attention RMSNorm (bit-identical; False for A/B)

    def _views(self, t: Optional[torch.Tensor], cut):
        """Slices of `t` along dim 0, the SAME view objects on every call for the same parent tensor at the same version: the
        host-flag cache matches tensors by identity, and a fresh view per call would cost a device read-back per slice per forward."""
        if t is None:
            return [None] * (len(cut) - 1)
        if _HostFlags._sig((t,)) is None:                  # no version counter (inference tensor): plain views, nothing cached
            return [t[cut[i]:cut[i + 1]] for i in range(len(cut) - 1)]
        import weakref
        for k in [k for k, e in self._slice_views.items() if e[0]() is None]:
            del self._slice_views[k]
        e = self._slice_views.get(id(t))
        if e is None or e[0]() is not t or e[1] != t._version or e[2] != cut:
            e = (weakref.ref(t), t._version, cut, [t[cut[i]:cut[i + 1]] for i in range(len(cut) - 1)])
            self._slice_views[id(t)] = e
        return e[3]

    def _forward_slices(self, parts: int, input_ids, attention_mask, pixel_values, pixel_attention_mask, position_ids, **kw):
        dev = self.w.device
        B = input_ids.shape[0]
        cur = torch.cuda.current_stream(dev)
        while len(self._side_streams) < parts:
            self._side_streams.append(torch.cuda.Stream(device=dev))
        cut = tuple(B * i // parts for i in range(parts + 1))
        ids, am, pv = self._views(input_ids, cut), self._views(attention_mask, cut), self._views(pixel_values, cut)
        pam, pos = self._views(pixel_attention_mask, cut), self._views(position_ids, cut)
        outs = []
        for i in range(parts):
            st = self._side_streams[i]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self._forward_one(ids[i], am[i], pv[i], pam[i], position_ids=pos[i], **kw))
        for i in range(parts):
            cur.wait_stream(self._side_streams[i])
        for o in outs:
            o.record_stream(cur)
        return torch.cat(outs, 0)

    # ----------------------------------------------------------------------------------- vision + connector
    def encode_images(self, pixel_values: torch.Tensor, pixel_attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(B, N, 3, H, W) [+ (B, N, H, W) bool] -> (n_real_images * r_latents, hidden) bf16
        (hf:idefics2/modeling_idefics2.py:817-862: all-zero padding images are dropped)."""
        a, w = self.arch, self.w
        dev = w.device
        B, N = pixel_values.shape[:2]
        P = a.v_patch
        pv = pixel_values.to(device=dev, dtype=torch.bfloat16).reshape(B * N, *pixel_values.shape[2:]).contiguous()
        Hh, Ww = pv.shape[-2:]
        gh, gw = Hh // P, Ww // P
        T = gh * gw
        pam = pixel_attention_mask.to(dev).reshape(B * N, Hh, Ww) if pixel_attention_mask is not None else None
        # padding-image flags, patch validity and NaViT position ids: one integer kernel on the device (csrc/frontend.hip)
        real, valid, pos_ids = frontend.idefics2_patch_front(pv, pam, P, a.v_image // P)
        flags = self._flags.get(pixel_values, pixel_attention_mask)
        if flags is None:                                  # first sight of these tensors: ONE 16-byte read-back
            both = torch.stack([real.sum(), (valid * real[:, None]).sum()]).cpu()
            flags = self._flags.put((int(both[0]), int(both[1])), pixel_values, pixel_attention_mask)
        n, n_valid = flags
        if n != B * N:                                     # all-zero padding images present: drop them (hf :831-836)
            keep = real.bool()
            pv, valid, pos_ids = pv[keep].contiguous(), valid[keep].contiguous(), pos_ids[keep].contiguous()
        all_valid = n_valid == n * T
        pos_ids = pos_ids.reshape(-1)

        E, nh, hd = a.v_hidden, a.v_heads, a.v_head_dim
        cols = ops.im2col_patches(pv, P, w.patch_ld)
        pos_rows = ops.embed_gather(pos_ids, w.pos, None, w.pos.shape[0])
        x = ops.linear(cols, w.patch_w, bias=w.patch_b, residual=pos_rows)
        del cols, pos_rows
        act = "gelu_tanh" if a.v_act == "gelu_pytorch_tanh" else "gelu"
        mode, kvld = (0, None) if all_valid else (2, valid)
        fp8v = w.fp8_vision and n * T >= 512            # e4m3 operands: the LayerNorms write the fp8 rows themselves, no bf16 copy
        for L in w.vit:
            if fp8v:
                qkv = ops.linear_fp8(*ops.layernorm_q8(x, L.ln1_w, L.ln1_b, a.v_ln_eps), *L.q8["qkv_w"], bias=L.qkv_b)
            else:
                qkv = ops.linear(ops.layernorm(x, L.ln1_w, L.ln1_b, a.v_ln_eps), L.qkv_w, bias=L.qkv_b)
            o = ops.attention(qkv, qkv.view(-1)[E:], qkv.view(-1)[2 * E:], n, T, T, nh, nh, hd, T * 3 * E, 3 * E, T * 3 * E, 3 * E,
                              hd ** -0.5, mode, key_valid=kvld)
            del qkv
            if fp8v:
                ops.linear_fp8(*ops.quantize_fp8(o.view(n * T, E)), *L.q8["out_w"], bias=L.out_b, residual=x, out=x)
                y = ops.linear_fp8(*ops.layernorm_q8(x, L.ln2_w, L.ln2_b, a.v_ln_eps), *L.q8["fc1_w"], bias=L.fc1_b, act=act)
                ops.linear_fp8(*ops.quantize_fp8(y), *L.q8["fc2_w"], bias=L.fc2_b, residual=x, out=x)
            else:
                ops.linear(o.view(n * T, E), L.out_w, bias=L.out_b, residual=x, out=x)
                y = ops.linear(ops.layernorm(x, L.ln2_w, L.ln2_b, a.v_ln_eps), L.fc1_w, bias=L.fc1_b, act=act)
                ops.linear(y, L.fc2_w, bias=L.fc2_b, residual=x, out=x)
            del y, o
        x = ops.layernorm(x, w.post_ln_w, w.post_ln_b, a.v_ln_eps)
        return self._connector(x, valid, n, T)

    def _connector(self, x: torch.Tensor, valid: torch.Tensor, n: int, T: int) -> torch.Tensor:
        """hf:idefics2/modeling_idefics2.py:757-760 (modality projection) + :708-743 (perceiver resampler, GQA)."""
        a, w = self.arch, self.w
        H, Lq, nh, nkv, hd = a.hidden_size, a.r_latents, a.r_heads, a.r_kv_heads, a.r_head_dim
        ctx = ops.linear(ops.linear(x, w.mp_gu, swiglu=True), w.mp_down)            # (n*T, H)
        lat = ops.tile_rows(w.latents, n * Lq)
        kvalid = torch.cat([valid, torch.ones((n, Lq), dtype=torch.int32, device=valid.device)], dim=1).contiguous()
        Sk = T + Lq
        kvin = torch.empty((n, Sk, H), dtype=torch.bfloat16, device=x.device)
        kdim = nkv * hd
        for P in w.perc:
            # both norms write straight into the concatenated [context ; latents] K/V input
            ops.rmsnorm(ctx, P.ctx_ln, a.rms_eps, 1, out=kvin, inner=T, ld_x=T * H, ld_out=Sk * H, rows=n * T, dim=H)
            ops.rmsnorm(lat, P.lat_ln, a.rms_eps, 1, out=kvin.view(-1)[T * H:], inner=Lq, ld_x=Lq * H, ld_out=Sk * H, rows=n * Lq, dim=H)
            latn = ops.rmsnorm(lat, P.lat_ln, a.rms_eps, 1)
            q = ops.linear(latn, P.q_w)
            kv = ops.linear(kvin.view(n * Sk, H), P.kv_w)
            o = ops.attention(q, kv, kv.view(-1)[kdim:], n, Lq, Sk, nh, nkv, hd, Lq * nh * hd, nh * hd, Sk * 2 * kdim, 2 * kdim,
                              hd ** -0.5, 2, key_valid=kvalid)
            ops.linear(o.view(n * Lq, nh * hd), P.o_w, residual=lat, out=lat)
            y = ops.rmsnorm(lat, P.post_ln, a.rms_eps, 1)
            ops.linear(ops.linear(y, P.gu_w, swiglu=True), P.down_w, residual=lat, out=lat)
        return ops.rmsnorm(lat, w.perc_ln, a.rms_eps, 1)

    # ----------------------------------------------------------------------------------- text side
    @staticmethod
    def _tlin(x: torch.Tensor, L: _TextLayer, name: str, **kw) -> torch.Tensor:
        """A text-stack projection: fp8 operands when the layer carries fp8 weights and the GEMM is large, else bf16."""
        if isinstance(x, tuple):                         # rows already in e4m3 (written by the norm kernel that produced them)
            return ops.linear_fp8(x[0], x[1], *L.q8[name], **kw)
        if L.q8 is not None and name in L.q8 and x.shape[0] >= 512:
            xq, xs = ops.quantize_fp8(x)
            wq, ws = L.q8[name]
            return ops.linear_fp8(xq, xs, wq, ws, **kw)
        return ops.linear(x, getattr(L, name), **kw)

    @staticmethod
    def _q8_in(L: _TextLayer, name: str, M: int) -> bool:
        """Does projection `name` of this layer take e4m3 rows (then its producer writes them, see ops.*_q8)?"""
        return L.q8 is not None and name in L.q8 and M >= 512

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                pixel_values: Optional[torch.Tensor] = None, pixel_attention_mask: Optional[torch.Tensor] = None,
                image_hidden_states: Optional[torch.Tensor] = None, icv: Optional[torch.Tensor] = None,
                hook_layers: Optional[Sequence[int]] = None, alpha: Optional[torch.Tensor] = None,
                capture: Optional[dict] = None, logits_rows: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, kv_cache: Optional[KVCache2] = None):
        """Returns logits (B, S, V) bf16.  icv (1, n_hooked, H) fp32 — already alpha-scaled when ``alpha`` is None;
        hook_layers: text-layer ids whose MLP OUTPUT (before the residual add) is edited."""
        parts = self.batch_streams
        B, S = input_ids.shape
        if (parts and parts > 1 and B >= 2 * parts and B * S >= parts * 2048 and capture is None and kv_cache is None
                and logits_rows is None and image_hidden_states is None and pixel_values is not None):
            return self._forward_slices(parts, input_ids, attention_mask, pixel_values, pixel_attention_mask, position_ids,
                                        icv=icv, hook_layers=hook_layers, alpha=alpha)
        return self._forward_one(input_ids, attention_mask, pixel_values, pixel_attention_mask, image_hidden_states, icv=icv,
                                 hook_layers=hook_layers, alpha=alpha, capture=capture, logits_rows=logits_rows,
                                 position_ids=position_ids, kv_cache=kv_cache)

    def _forward_one(self, input_ids, attention_mask=None, pixel_values=None, pixel_attention_mask=None, image_hidden_states=None,
                     icv=None, hook_layers=None, alpha=None, capture=None, logits_rows=None, position_ids=None, kv_cache=None):
        a, w = self.arch, self.w
        dev = w.device
        B, S = input_ids.shape
        # `capture` records per-layer tensors and, by default, takes the unfused kernels (every intermediate exists as a tensor);
        # with self.capture_keeps_path the product's fused path runs and only the layer outputs are recorded (diagnostics)
        cap = None if getattr(self, "capture_keeps_path", False) else capture
        past = kv_cache.len if kv_cache is not None else 0
        Sk = past + S
        assert Sk <= w.max_positions, "sequence longer than the rotary table"
        M, H, nh, nkv, hd = B * S, a.hidden_size, a.num_heads, a.num_kv_heads, a.head_dim
        if attention_mask is None:
            attention_mask = torch.ones((B, Sk), dtype=torch.long, device=dev)
        assert attention_mask.shape[1] == Sk, "attention_mask must span past + new tokens"
        ids = input_ids.to(dev).contiguous()
        if image_hidden_states is None and pixel_values is not None:
            image_hidden_states = self.encode_images(pixel_values, pixel_attention_mask)
        h = ops.embed_gather(ids, w.embed, None, w.embed.shape[0]).view(M, H)
        if image_hidden_states is not None:
            # inputs_merger on the device: rank of every <image> token + row copy (no nonzero(), no host round trip); the token
            # count is validated against the rows at hand once per distinct input_ids tensor
            img = image_hidden_states.reshape(-1, H).contiguous()
            count = frontend.merge_image_rows_(h, ids, img, a.image_token_id)
            n_tok = self._flags.get(input_ids)
            if n_tok is None:
                n_tok = self._flags.put(int(count), input_ids)
            if n_tok != img.shape[0]:
                raise ValueError(f"{n_tok} <image> tokens in input_ids but {img.shape[0]} image hidden states")
        key_valid = attention_mask.to(device=dev, dtype=torch.int32).contiguous()
        if position_ids is None:                        # plain forward: arange (MistralModel.forward); generate passes mask-derived ids
            pos = torch.arange(past, Sk, device=dev, dtype=torch.int64).repeat(B).contiguous()
        else:
            pos = position_ids.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
            assert pos.numel() == M
        idx_of = {int(l): i for i, l in enumerate(hook_layers)} if (icv is not None and hook_layers is not None) else {}
        if icv is not None:
            icv = icv.to(device=dev, dtype=torch.float32).contiguous()
            if alpha is not None:
                alpha = alpha.to(device=dev, dtype=torch.float32).contiguous()
        qd, kd = nh * hd, nkv * hd
        ldq = qd + 2 * kd
        xn = None
        for l, L in enumerate(w.text):
            if xn is not None:
                x = xn
            elif self._q8_in(L, "qkv_w", M) and cap is None:
                x = ops.rmsnorm_q8(h, L.in_ln, a.rms_eps, 1)
            else:
                x = ops.rmsnorm(h, L.in_ln, a.rms_eps, 1)
            xn = None
            if kv_cache is not None and S == 1:                       # a decode step: rotary + append + GQA attention in one launch
                qs = ops.linear_produce(x, L.qkv_w) if not isinstance(x, tuple) else None
                o = ops.decode_attn(qs if qs is not None else self._tlin(x, L, "qkv_w"), w.cos, w.sin, pos, kv_cache.kv[l], past, nh, nkv, hd,
                                    hd ** -0.5, key_valid=key_valid, kv_rows=kv_cache.rows)
                qkv = None
            else:
                qkv = self._tlin(x, L, "qkv_w")
                ops.rotary_(qkv, w.cos, w.sin, pos, M, nh + nkv, hd, ldq, 0, 1)   # Q heads | K heads are contiguous in the fused row: one launch
            if qkv is None:
                pass
            elif kv_cache is None:
                o = ops.attention(qkv, qkv.view(-1)[qd:], qkv.view(-1)[qd + kd:], B, S, S, nh, nkv, hd, S * ldq, ldq, S * ldq, ldq,
                                  hd ** -0.5, 1, key_valid=key_valid)
            else:
                cache = kv_cache.kv[l]
                cache[:, past:Sk] = qkv.view(B, S, ldq)[:, :, qd:]                         # append K|V (device copy)
                o = ops.attention(qkv, cache, cache.view(-1)[kd:], B, S, Sk, nh, nkv, hd, S * ldq, ldq, kv_cache.max_len * 2 * kd, 2 * kd,
                                  hd ** -0.5, 1, key_valid=key_valid)
            if M >= 512 and cap is None and self.fold_residual:   # the residual add folded into the norm that follows (as in IdeficsEngine): bit-identical
                if self._q8_in(L, "gu_w", M):
                    x = ops.add_rmsnorm_q8_(h, self._tlin(o.view(M, qd), L, "o_w"), L.post_ln, a.rms_eps, 1)
                else:
                    x = ops.add_rmsnorm_(h, self._tlin(o.view(M, qd), L, "o_w"), L.post_ln, a.rms_eps, 1)
            else:
                self._tlin(o.view(M, qd), L, "o_w", residual=h, out=h)
                x = ops.rmsnorm(h, L.post_ln, a.rms_eps, 1)
            act = self._tlin(x, L, "gu_w", swiglu=True)
            del qkv, o, x
            if l in idx_of:
                i = idx_of[l]
                m = self._tlin(act, L, "down_w")                                           # raw MLP branch (bf16)
                if cap is not None:
                    capture.setdefault("mlp_raw", []).append(m.view(B, S, H).clone())
                al = alpha[0, i:i + 1] if alpha is not None else None
                nw = w.text[l + 1].in_ln if l + 1 < a.num_layers else w.final_ln
                if self.fuse_hook_norm and cap is None and l + 1 < a.num_layers and self._q8_in(w.text[l + 1], "qkv_w", M):
                    h, xq_, xs_ = ops.inject_renorm_add_q8(m, icv[0, i], h, al, nw, norm_eps=a.rms_eps, norm_flavour=1)
                    xn = (xq_, xs_)
                elif self.fuse_hook_norm:
                    h, xn = ops.inject_renorm_add(m, icv[0, i], h, alpha=al, norm_weight=nw, norm_eps=a.rms_eps, norm_flavour=1)
                else:
                    h = ops.inject_renorm_add(m, icv[0, i], h, alpha=al)
                del m
            else:
                if cap is not None:
                    capture.setdefault("mlp_raw", []).append(ops.linear(act, L.down_w).view(B, S, H).clone())
                self._tlin(act, L, "down_w", residual=h, out=h)
            del act
            if capture is not None:
                capture.setdefault("layer_out", []).append(h.view(B, S, H).clone())
        if kv_cache is not None:
            kv_cache.len = Sk
        x = xn if xn is not None else ops.rmsnorm(h, w.final_ln, a.rms_eps, 1)
        if capture is not None:
            capture["image_hidden_states"] = image_hidden_states
        if logits_rows is not None:
            return ops.linear(x.index_select(0, logits_rows), w.lm_head)
        logits = ops.linear(x, w.lm_head)
        V = logits.shape[-1]
        return logits.view(B, S, V) if logits.is_contiguous() else logits.as_strided((B, S, V), (S * logits.stride(0), logits.stride(0), 1))
