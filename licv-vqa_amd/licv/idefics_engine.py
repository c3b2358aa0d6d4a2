"""Native Idefics forward for MI355X with the ICV hook fused into the layer loop.

Replaces, for the hot path, ``IdeficsForVisionText2Text.forward`` (hf:idefics/modeling_idefics.py:934-1084,
:1165-1182) + the baukit hook mechanism (ref:icv_src/icv_model/icv_intervention.py:88-98).  Every
arithmetic step is a HIP kernel reached through the C-ABI (``licv.ops``); torch supplies device memory,
the stream and a handful of integer mask preparations (position ids, key-valid mask, gate).

Data layout in HBM (all row-major, tokens flattened to M = B*S rows):
  * residual stream ``h``: (M, H); bf16 until the first hooked layer, fp32 afterwards (the fp32 ICV
    promotes it in the reference) — every later residual add happens in fp32;
  * GEMM inputs/outputs bf16; fused projections: QKV (3H x H), cross-attn KV (2H x E_v), gate|up
    interleaved in 16-row blocks for the SwiGLU epilogue, LM head + additional_fc concatenated;
  * attention reads Q/K/V straight out of the fused projection buffers through strides (no transposes);
  * vision tokens (B*N_img*257, 1280) -> perceiver latents (B, N_img*64, 1280) -> cross-attn K/V.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .config import IdeficsArch


def _bf(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.bfloat16).contiguous()


def _pad_cols(w: torch.Tensor, to: int) -> torch.Tensor:
    if w.shape[1] == to:
        return w.contiguous()
    out = torch.zeros((w.shape[0], to), dtype=w.dtype, device=w.device)
    out[:, : w.shape[1]] = w
    return out


@dataclass
class _VitLayer:
    ln1_w: torch.Tensor; ln1_b: torch.Tensor; qkv_w: torch.Tensor; qkv_b: torch.Tensor
    out_w: torch.Tensor; out_b: torch.Tensor; ln2_w: torch.Tensor; ln2_b: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor; fc2_w: torch.Tensor; fc2_b: torch.Tensor


@dataclass
class _PercBlock:
    ctx_w: torch.Tensor; ctx_b: torch.Tensor; lat_w: torch.Tensor; lat_b: torch.Tensor
    q_w: torch.Tensor; kv_w: torch.Tensor; out_w: torch.Tensor
    qn_w: Optional[torch.Tensor]; qn_b: Optional[torch.Tensor]; kn_w: Optional[torch.Tensor]; kn_b: Optional[torch.Tensor]
    ln_w: torch.Tensor; ln_b: torch.Tensor; fc_w: torch.Tensor; cproj_w: torch.Tensor


@dataclass
class _DecLayer:
    in_ln: torch.Tensor; qkv_w: torch.Tensor; o_w: torch.Tensor; post_ln: torch.Tensor
    gu_w: torch.Tensor; down_w: torch.Tensor


@dataclass
class _XLayer:
    in_ln: torch.Tensor; q_w: torch.Tensor; kv_w: torch.Tensor; o_w: torch.Tensor
    qn_w: Optional[torch.Tensor]; kn_w: Optional[torch.Tensor]
    post_ln: torch.Tensor; gu_w: torch.Tensor; down_w: torch.Tensor
    gate_attn: float; gate_dense: float


class IdeficsWeights:
    """Engine-layout weights built from an HF-named state dict (bf16)."""

    def __init__(self, sd: Dict[str, torch.Tensor], arch: IdeficsArch, device="cuda"):
        a, dev = arch, torch.device(device)
        self.arch, self.device = arch, dev
        g = lambda k: _bf(sd[k], dev)
        self.embed = g("model.embed_tokens.weight")
        self.embed_extra = g("model.embed_tokens.additional_embedding.weight") if a.additional_vocab_size else None
        vp = "model.vision_model."
        kdim = 3 * a.v_patch * a.v_patch
        self.patch_ld = (kdim + 63) // 64 * 64
        self.patch_w = _pad_cols(g(vp + "embeddings.patch_embedding.weight").flatten(1), self.patch_ld)
        self.cls = g(vp + "embeddings.class_embedding")
        self.pos = g(vp + "embeddings.position_embedding.weight")
        self.pre_ln_w, self.pre_ln_b = g(vp + "pre_layrnorm.weight"), g(vp + "pre_layrnorm.bias")
        self.vit: List[_VitLayer] = []
        for i in range(a.v_layers):
            p = f"{vp}encoder.layers.{i}."
            cat = lambda suf: torch.cat([g(p + f"self_attn.{n}_proj.{suf}") for n in ("q", "k", "v")]).contiguous()
            self.vit.append(_VitLayer(g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"), cat("weight"), cat("bias"),
                                      g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"),
                                      g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"),
                                      g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"), g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias")))
        self.perc: List[_PercBlock] = []
        if a.use_resampler:
            rp = "model.perceiver_resampler."
            self.latents = g(rp + "latents")
            for i in range(a.r_depth):
                p, m = f"{rp}blocks.{i}.0.", f"{rp}blocks.{i}.1."
                qn = a.r_qk_norm
                self.perc.append(_PercBlock(
                    g(p + "context_layer_norm.weight"), g(p + "context_layer_norm.bias"),
                    g(p + "latents_layer_norm.weight"), g(p + "latents_layer_norm.bias"),
                    g(p + "q_proj.weight"), torch.cat([g(p + "k_proj.weight"), g(p + "v_proj.weight")]).contiguous(),
                    g(p + "output_proj.weight"),
                    g(p + "q_layer_norm.weight") if qn else None, g(p + "q_layer_norm.bias") if qn else None,
                    g(p + "k_layer_norm.weight") if qn else None, g(p + "k_layer_norm.bias") if qn else None,
                    g(m + "ln.weight"), g(m + "ln.bias"), g(m + "fc.weight"), g(m + "c_proj.weight")))
            self.perc_ln_w, self.perc_ln_b = g(rp + "layer_norm.weight"), g(rp + "layer_norm.bias")
        self.dec: List[_DecLayer] = []
        for i in range(a.num_layers):
            p = f"model.layers.{i}."
            qkv = torch.cat([g(p + f"self_attn.{n}_proj.weight") for n in ("q", "k", "v")]).contiguous()
            gu = ops.pack_gate_up(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight"))
            self.dec.append(_DecLayer(g(p + "input_layernorm.weight"), qkv, g(p + "self_attn.o_proj.weight"),
                                      g(p + "post_attention_layernorm.weight"), gu, g(p + "mlp.down_proj.weight")))
        self.xat: List[_XLayer] = []
        for j in range(a.num_cross_layers):
            p = f"model.gated_cross_attn_layers.{j}."
            kv = torch.cat([g(p + "cross_attn.k_proj.weight"), g(p + "cross_attn.v_proj.weight")]).contiguous()
            gu = ops.pack_gate_up(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight"))
            # tanh(alpha) evaluated in bf16 like the module does (hf:idefics/modeling_idefics.py:793,800)
            ga = float(torch.tanh(g(p + "alpha_cross_attn")).float().reshape(-1)[0])
            gd = float(torch.tanh(g(p + "alpha_dense")).float().reshape(-1)[0])
            self.xat.append(_XLayer(g(p + "input_layernorm.weight"), g(p + "cross_attn.q_proj.weight"), kv,
                                    g(p + "cross_attn.o_proj.weight"),
                                    g(p + "cross_attn.q_layer_norm.weight") if a.qk_layer_norms else None,
                                    g(p + "cross_attn.k_layer_norm.weight") if a.qk_layer_norms else None,
                                    g(p + "post_attention_layernorm.weight"), gu, g(p + "mlp.down_proj.weight"), ga, gd))
        self.final_ln = g("model.norm.weight")
        head = g("lm_head.weight")
        if a.additional_vocab_size:
            head = torch.cat([head, g("lm_head.additional_fc.weight")]).contiguous()
        self.lm_head = head
        # rotary tables exactly as hf:idefics/modeling_idefics.py:357-379 builds them (fp32 maths, cast to bf16)
        inv = 1.0 / (a.rope_base ** (torch.arange(0, a.head_dim, 2, dtype=torch.int64).to(torch.float) / a.head_dim))
        t = torch.arange(a.max_positions, dtype=torch.int64).to(torch.float)
        emb = torch.cat((torch.outer(t, inv),) * 2, dim=-1)
        self.cos, self.sin = _bf(emb.cos(), dev), _bf(emb.sin(), dev)


class KVCache:
    """Per-layer (rows, max_len, [K | V]) bf16 self-attention cache for hooked generate, ONE allocation for all layers.

    Beam search never moves it.  The cache is allocated with batch x beams rows; the prefill fills the first `batch` rows (question b's
    prompt in row b); `replicate` only builds the row table `rows` (rows x max_len int32: the physical row that holds position p of
    beam row r's history); a decode step appends row r's new token to physical row r (positions >= prompt length, so prompts and
    appended tokens never collide) and reads its history through the table (licv_decode_attn); a beam reorder permutes table rows
    (licv_beam_step writes the new table itself; `reorder` is the torch form of the same update for callers that only have the
    source-beam indices).  The gather of the whole cache per step - and its transient second copy - is gone."""

    def __init__(self, arch, batch: int, max_len: int, device, beams: int = 1, width: Optional[int] = None):
        width = 2 * arch.hidden_size if width is None else width
        self.max_len, self.len, self.batch, self.beams = max_len, 0, batch, beams
        self._all = torch.empty((arch.num_layers, batch * beams, max_len, width), dtype=torch.bfloat16, device=device)
        self.kv = [t[:batch] for t in self._all.unbind(0)]          # the prefill's view: the first `batch` rows of every layer
        self.rows = None       # (batch * beams, max_len) int32 after replicate(); None: every row reads its own cache row
        self.xkv = None        # cross-attention K|V per gated layer, projected once at the prefill by the native runner (step-invariant)

    def replicate(self, nb: int):
        """The prompt state of every question is shared by its `nb` beams: no copy, only the row table."""
        assert nb == self.beams, f"the cache was allocated for {self.beams} beams per question, not {nb}"
        n = self.batch * nb
        dev = self._all.device
        own = torch.arange(n, device=dev, dtype=torch.int32)
        self.rows = own.unsqueeze(1).repeat(1, self.max_len)
        self.rows[:, : self.len] = (own // nb).unsqueeze(1)
        self.kv = list(self._all.unbind(0))

    def reorder(self, idx: torch.Tensor):
        """Beam r continues the history of beam idx[r]; its next token goes to its own row."""
        rows = self.rows.index_select(0, idx)
        rows[:, self.len] = torch.arange(rows.shape[0], device=rows.device, dtype=torch.int32)
        self.rows = rows

    def set_rows(self, rows: torch.Tensor):
        """The table as licv_beam_step left it (same content as reorder(), no launch)."""
        self.rows = rows


class IdeficsEngine:
    def __init__(self, weights: IdeficsWeights, fuse_hook_norm: bool = True, use_runner: bool = True):
        self.w, self.arch = weights, weights.arch
        self.fuse_hook_norm = fuse_hook_norm
        # the language stack through ONE C call (csrc/runner.hip) whenever nothing is captured: same kernels, same results, no
        # per-launch interpreter cost (decode steps and 32-token passes are launch-bound from Python)
        self.use_runner = use_runner
        self._runner = None
        # A batch of questions is cut in `batch_streams` slices that run on HIP streams of their own: the rows of a batch are
        # independent (property P2), the kernels and the results are the same, and the last, partly filled round of one slice's
        # GEMM (N = 4096 outputs are 400 tiles on 256 CUs: 1.56 rounds) is filled by the other slice's workgroups.
        # Measured at the headline shape: 222.9 -> 213.8 ms with 2 slices (3 / 4 slices: slower).  0 / 1 = off.
        self.batch_streams = 2
        self._side_streams = []

    def _forward_slices(self, parts: int, input_ids, attention_mask, pixel_values, image_attention_mask, image_states,
                        logits_rows=None, **kw):
        dev = self.w.device
        B, S = input_ids.shape
        cur = torch.cuda.current_stream(dev)
        cut = [B * i // parts for i in range(parts + 1)]
        rows_of = [None] * parts
        if logits_rows is not None:
            # selected rows (flattened b * S + s, ascending — the trainer's answer rows): slice i takes the rows of its questions,
            # re-based; the split points come back in ONE small read (the callers that pass rows have just synchronised to build them)
            r = logits_rows.reshape(-1)
            bounds = torch.tensor([c * S for c in cut], device=dev, dtype=r.dtype)
            info = torch.cat([torch.searchsorted(r, bounds), (r[1:] < r[:-1]).any().reshape(1).to(bounds.dtype)]).tolist()
            if info[-1] or r.numel() == 0:                            # not ascending (or nothing selected): one plain pass
                return self._forward_one(input_ids, attention_mask, pixel_values, image_attention_mask, image_states,
                                         logits_rows=logits_rows, **kw)
            rows_of = [(r[info[i]:info[i + 1]] - cut[i] * S).contiguous() for i in range(parts)]
        while len(self._side_streams) < parts:
            self._side_streams.append(torch.cuda.Stream(device=dev))
        outs = []
        for i in range(parts):
            if logits_rows is not None and rows_of[i].numel() == 0:
                continue
            st = self._side_streams[i]
            st.wait_stream(cur)                                       # the inputs were produced on the caller's stream
            sl = slice(cut[i], cut[i + 1])
            with torch.cuda.stream(st):
                outs.append(self._forward_one(input_ids[sl], None if attention_mask is None else attention_mask[sl],
                                              None if pixel_values is None else pixel_values[sl], image_attention_mask[sl],
                                              None if image_states is None else image_states[sl], logits_rows=rows_of[i], **kw))
        for i in range(parts):
            cur.wait_stream(self._side_streams[i])
        for o in outs:
            o.record_stream(cur)                                      # allocated on a side stream, read on the caller's from here on
        return torch.cat(outs, 0)

    # ----------------------------------------------------------------------------------- vision side
    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values (B, N, 3, H, W) -> image_hidden_states (B, N*image_seq_len, E) bf16
        (hf:idefics/modeling_idefics.py:986-1013; vision.py:341-381; perceiver.py:93-103)."""
        a, w = self.arch, self.w
        B, N = pixel_values.shape[:2]
        n_img = B * N
        pix = pixel_values.to(torch.bfloat16).reshape(n_img, *pixel_values.shape[2:]).contiguous()
        assert pix.shape[-1] == a.v_image and pix.shape[-2] == a.v_image, "image size must match the vision config"
        E, T, nh, hd = a.v_embed, a.v_tokens, a.v_heads, a.v_head_dim
        cols = ops.im2col_patches(pix, a.v_patch, w.patch_ld)
        patches = ops.linear(cols, w.patch_w)
        del cols
        x = ops.vit_embed_ln(patches, w.cls, w.pos, w.pre_ln_w, w.pre_ln_b, n_img, T - 1, a.v_ln_eps).view(n_img * T, E)
        del patches
        act = "gelu" if a.v_act == "gelu" else "gelu_tanh"
        for L in w.vit:
            y = ops.layernorm(x, L.ln1_w, L.ln1_b, a.v_ln_eps)
            qkv = ops.linear(y, L.qkv_w, bias=L.qkv_b)
            o = ops.attention(qkv, qkv.view(-1)[E:], qkv.view(-1)[2 * E:], n_img, T, T, nh, nh, hd,
                              T * 3 * E, 3 * E, T * 3 * E, 3 * E, hd ** -0.5, 0)
            del qkv
            ops.linear(o.view(n_img * T, E), L.out_w, bias=L.out_b, residual=x, out=x)
            y = ops.layernorm(x, L.ln2_w, L.ln2_b, a.v_ln_eps)
            y = ops.linear(y, L.fc1_w, bias=L.fc1_b, act=act)
            ops.linear(y, L.fc2_w, bias=L.fc2_b, residual=x, out=x)
            del y, o
        if not a.use_resampler:
            return x.view(B, N * T, E)
        return self._perceiver(x, n_img).view(B, N * a.r_latents, E)

    def _perceiver(self, ctx: torch.Tensor, n_img: int) -> torch.Tensor:
        a, w = self.arch, self.w
        E, T, Lq, nh, hd = a.v_embed, a.v_tokens, a.r_latents, a.r_heads, a.r_head_dim
        inner = nh * hd
        lat = ops.tile_rows(w.latents, n_img * Lq)
        kvin = torch.empty((n_img, T + Lq, E), dtype=torch.bfloat16, device=ctx.device)
        for P in w.perc:
            ops.layernorm(ctx, P.ctx_w, P.ctx_b, 1e-5, out=kvin, out_group=T, out_group_extra=Lq * E)
            ops.layernorm(lat, P.lat_w, P.lat_b, 1e-5, out=kvin.view(-1)[T * E:], out_group=Lq, out_group_extra=T * E)
            latn = ops.layernorm(lat, P.lat_w, P.lat_b, 1e-5)
            q = ops.linear(latn, P.q_w)
            kv = ops.linear(kvin.view(n_img * (T + Lq), E), P.kv_w)
            if P.qn_w is not None:
                ops.layernorm(q, P.qn_w, P.qn_b, 1e-5, out=q, inner=nh, ld_x=inner, ld_out=inner, rows=n_img * Lq * nh, dim=hd)
                ops.layernorm(kv, P.kn_w, P.kn_b, 1e-5, out=kv, inner=nh, ld_x=2 * inner, ld_out=2 * inner,
                              rows=n_img * (T + Lq) * nh, dim=hd)
            o = ops.attention(q, kv, kv.view(-1)[inner:], n_img, Lq, T + Lq, nh, nh, hd, Lq * inner, inner,
                              (T + Lq) * 2 * inner, 2 * inner, hd ** -0.5, 0)
            ops.linear(o.view(n_img * Lq, inner), P.out_w, residual=lat, out=lat)
            y = ops.layernorm(lat, P.ln_w, P.ln_b, 1e-5)
            y = ops.linear(y, P.fc_w, act="relu")
            ops.linear(y, P.cproj_w, residual=lat, out=lat)
        return ops.layernorm(lat, w.perc_ln_w, w.perc_ln_b, 1e-5)

    # ----------------------------------------------------------------------------------- language side
    @staticmethod
    def _position_ids(attention_mask: torch.Tensor, q_len: int) -> torch.Tensor:
        pos = attention_mask.long().cumsum(-1) - 1                       # hf:idefics/modeling_idefics.py:972-976
        pos = pos.masked_fill(attention_mask == 0, 1)
        return pos[:, -q_len:].contiguous()

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                pixel_values: Optional[torch.Tensor] = None, image_attention_mask: Optional[torch.Tensor] = None,
                image_states: Optional[torch.Tensor] = None, icv: Optional[torch.Tensor] = None,
                hook_layers: Optional[Sequence[int]] = None, alpha: Optional[torch.Tensor] = None,
                capture: Optional[dict] = None, kv_cache: Optional[KVCache] = None,
                save_hook_inputs: Optional[list] = None, logits_rows: Optional[torch.Tensor] = None):
        """Returns logits (B, S, V) bf16 (a view with a padded row stride).
        icv: (1, n_hooked, H) fp32 — already alpha-scaled when ``alpha`` is None (the reference contract,
        ref:icv_src/icv_module.py:89-92); with ``alpha`` (1, n_hooked) fp32 the scaling is folded into the kernel.
        hook_layers: decoder-layer ids whose OUTPUT is edited (ref:config/lmm/idefics-9B.yaml:7)."""
        parts = self.batch_streams
        B, S = input_ids.shape
        if (parts and parts > 1 and B >= 2 * parts and B * S >= parts * 2048 and capture is None and kv_cache is None
                and save_hook_inputs is None and image_attention_mask is not None):
            return self._forward_slices(parts, input_ids, attention_mask, pixel_values, image_attention_mask, image_states,
                                        logits_rows=logits_rows, icv=icv, hook_layers=hook_layers, alpha=alpha)
        return self._forward_one(input_ids, attention_mask, pixel_values, image_attention_mask, image_states, icv=icv,
                                 hook_layers=hook_layers, alpha=alpha, capture=capture, kv_cache=kv_cache,
                                 save_hook_inputs=save_hook_inputs, logits_rows=logits_rows)

    def _forward_one(self, input_ids, attention_mask=None, pixel_values=None, image_attention_mask=None, image_states=None, icv=None,
                     hook_layers=None, alpha=None, capture=None, kv_cache=None, save_hook_inputs=None, logits_rows=None):
        a, w = self.arch, self.w
        dev = w.device
        B, S = input_ids.shape
        M, H, nh, hd = B * S, a.hidden_size, a.num_heads, a.head_dim
        if attention_mask is None:
            attention_mask = torch.ones((B, S + (kv_cache.len if kv_cache else 0)), dtype=torch.long, device=dev)
        Sk = attention_mask.shape[1]
        if image_states is None:
            image_states = self.encode_images(pixel_values)
        Nk = image_states.shape[1]
        E = image_states.shape[2]
        img_len = a.image_seq_len
        n_img = Nk // img_len
        img_mask = image_attention_mask.to(torch.int32).contiguous()
        gate = (img_mask != 0).any(-1).to(torch.float32).reshape(-1).contiguous()      # cross_attention_gate
        key_valid = attention_mask.to(torch.int32).contiguous()
        pos = self._position_ids(attention_mask, S).reshape(-1)
        idx_of = {int(l): i for i, l in enumerate(hook_layers)} if (icv is not None and hook_layers is not None) else {}
        if icv is not None:
            icv = icv.to(device=dev, dtype=torch.float32).contiguous()
            if alpha is not None:
                alpha = alpha.to(device=dev, dtype=torch.float32).contiguous()

        past = kv_cache.len if kv_cache is not None else 0
        assert past + S == Sk, "attention_mask must span past + new tokens"
        if self.use_runner and self.fuse_hook_norm and capture is None and save_hook_inputs is None and not ops.profiling():
            if self._runner is None:
                from .runner import TextRunner
                self._runner = TextRunner(w)
            return self._runner.forward(input_ids.contiguous(), key_valid, pos, image_states.contiguous(), img_mask, gate,
                                        icv=icv, alpha=alpha.reshape(-1).contiguous() if alpha is not None else None,
                                        hook_layers=hook_layers if idx_of else None, kv_cache=kv_cache, logits_rows=logits_rows)
        if kv_cache is not None and kv_cache.xkv is not None:
            kv_cache.xkv = None                      # the Python loop re-projects the cross-attention K|V at every step
        h = ops.embed_gather(input_ids.contiguous(), w.embed, w.embed_extra, a.vocab_size).view(M, H)
        xn = None                                    # RMSNorm of h for the next block, when the hook kernel made it
        img2d = image_states.reshape(B * Nk, E)

        def next_norm_weight(l: int):
            if l + 1 >= a.num_layers:
                return w.final_ln
            if (l + 1) % a.cross_layer_interval == 0:
                return w.xat[(l + 1) // a.cross_layer_interval].in_ln
            return w.dec[l + 1].in_ln

        foldable = M >= 512 and capture is None and save_hook_inputs is None and self.fuse_hook_norm
        pending = None                               # a cross layer's MLP branch (and its gate scale) waiting for the next norm to add it
        for l in range(a.num_layers):
            if l % a.cross_layer_interval == 0:
                X = w.xat[l // a.cross_layer_interval]
                x = xn if xn is not None else ops.rmsnorm(h, X.in_ln, a.rms_eps)
                xn = None
                q = ops.linear(x, X.q_w)
                kv = ops.linear(img2d, X.kv_w)
                if X.qn_w is not None:
                    ops.rmsnorm(q, X.qn_w, a.rms_eps, out=q, inner=nh, ld_x=H, ld_out=H, rows=M * nh, dim=hd)
                    ops.rmsnorm(kv, X.kn_w, a.rms_eps, out=kv, inner=nh, ld_x=2 * H, ld_out=2 * H, rows=B * Nk * nh, dim=hd)
                o = ops.attention(q, kv, kv.view(-1)[H:], B, S, Nk, nh, nh, hd, S * H, H, Nk * 2 * H, 2 * H, hd ** -0.5, 3,
                                  img_mask=img_mask, img_len=img_len)
                if foldable:                                # both gated residual adds folded into the norms that follow (see below)
                    x = ops.add_rmsnorm_(h, ops.linear(o.view(M, H), X.o_w), X.post_ln, a.rms_eps, row_gate=gate, scale=X.gate_attn)
                    act = ops.linear(x, X.gu_w, swiglu=True)
                    pending = (ops.linear(act, X.down_w), X.gate_dense)     # added by the decoder layer's input norm
                else:
                    ops.linear(o.view(M, H), X.o_w, row_gate=gate, scale=X.gate_attn, residual=h, out=h)
                    x = ops.rmsnorm(h, X.post_ln, a.rms_eps)
                    act = ops.linear(x, X.gu_w, swiglu=True)
                    ops.linear(act, X.down_w, scale=X.gate_dense, residual=h, out=h)
                del q, kv, o, act
            D = w.dec[l]
            if pending is not None:
                x = ops.add_rmsnorm_(h, pending[0], D.in_ln, a.rms_eps, scale=pending[1])
                pending = None
            else:
                x = xn if xn is not None else ops.rmsnorm(h, D.in_ln, a.rms_eps)
            xn = None
            if kv_cache is None:
                qkv = ops.linear(x, D.qkv_w)
                ops.rotary_(qkv, w.cos, w.sin, pos, M, nh, hd, 3 * H, H, 2)
                o = ops.attention(qkv, qkv.view(-1)[H:], qkv.view(-1)[2 * H:], B, S, S, nh, nh, hd, S * 3 * H, 3 * H,
                                  S * 3 * H, 3 * H, hd ** -0.5, 1, key_valid=key_valid)
            elif S == 1:                                                 # a decode step: rotary + append + attention in one launch
                qkv = None
                qs = ops.linear_produce(x, D.qkv_w)
                o = ops.decode_attn(qs if qs is not None else ops.linear(x, D.qkv_w), w.cos, w.sin, pos, kv_cache.kv[l], past, nh, nh, hd,
                                    hd ** -0.5, key_valid=key_valid, kv_rows=kv_cache.rows)
            else:
                qkv = ops.linear(x, D.qkv_w)
                ops.rotary_(qkv, w.cos, w.sin, pos, M, nh, hd, 3 * H, H, 2)
                cache = kv_cache.kv[l]
                cache[:, past:past + S] = qkv.view(B, S, 3 * H)[:, :, H:]          # append K|V (device copy)
                o = ops.attention(qkv, cache, cache.view(-1)[H:], B, S, Sk, nh, nh, hd, S * 3 * H, 3 * H,
                                  kv_cache.max_len * 2 * H, 2 * H, hd ** -0.5, 1, key_valid=key_valid)
            # Large batches (the 256-tile GEMMs), nothing captured: the layer's two residual adds leave the GEMM epilogues (a
            # read-modify-write of the fp32 stream costs the o / down projections 16-28 %) and are folded into the row kernels that
            # follow — same sums, same rounding points, bit-identical — so both projections take the register-direct epilogue
            fold = foldable
            if fold:
                x = ops.add_rmsnorm_(h, ops.linear(o.view(M, H), D.o_w), D.post_ln, a.rms_eps)
            else:
                ops.linear(o.view(M, H), D.o_w, residual=h, out=h)
                x = ops.rmsnorm(h, D.post_ln, a.rms_eps)
            act = ops.linear(x, D.gu_w, swiglu=True)
            if fold and l in idx_of:
                i = idx_of[l]
                br = ops.linear(act, D.down_w)
                del qkv, o, act
                h, xn = ops.inject_renorm(h, icv[0, i], alpha=alpha[0, i:i + 1] if alpha is not None else None,
                                          out=h if h.dtype == torch.float32 else None, norm_weight=next_norm_weight(l),
                                          norm_eps=a.rms_eps, pre=br)
                del br
                continue
            ops.linear(act, D.down_w, residual=h, out=h)
            del qkv, o, act
            if capture is not None:
                capture.setdefault("raw", []).append(h.view(B, S, H).clone())
            if l in idx_of:
                i = idx_of[l]
                if save_hook_inputs is not None:
                    save_hook_inputs.append(h if h.dtype == torch.float32 else h.clone())
                al = alpha[0, i:i + 1] if alpha is not None else None
                in_place = h.dtype == torch.float32 and save_hook_inputs is None
                if self.fuse_hook_norm:
                    h, xn = ops.inject_renorm(h, icv[0, i], alpha=al, out=h if in_place else None,
                                              norm_weight=next_norm_weight(l), norm_eps=a.rms_eps)
                else:
                    h = ops.inject_renorm(h, icv[0, i], alpha=al, out=h if in_place else None)
            if capture is not None:
                capture.setdefault("edited", []).append(h.view(B, S, H).clone())
        if kv_cache is not None:
            kv_cache.len = past + S
        x = xn if xn is not None else ops.rmsnorm(h, w.final_ln, a.rms_eps)
        if capture is not None:
            capture["final_norm"] = x.view(B, S, H).clone()
            capture["image_states"] = image_states
        if logits_rows is not None:
            x = x.index_select(0, logits_rows)
            return ops.linear(x, w.lm_head)
        logits = ops.linear(x, w.lm_head)
        return logits.view(B, S, logits.shape[-1]) if logits.is_contiguous() else \
            logits.as_strided((B, S, logits.shape[-1]), (S * logits.stride(0), logits.stride(0), 1))
