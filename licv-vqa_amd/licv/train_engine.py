"""Student pass WITH gradient for L-ICV training (ref:icv_src/icv_module.py:71-119).

Only ``icv`` and ``alpha`` are trainable and the LMM is frozen, so the backward needs exactly one thing:
d loss / d hidden-state propagated from the masked-KL rows down through the language stack, with the hook's
backward kernel peeling off d loss / d (alpha*icv) at every hooked layer.  The vision side, the embeddings
and the cross-attention K/V never receive gradient (they do not depend on the ICV).

Forward = the inference engine's layer loop with the MLP gate/up left unfused so the backward has g and u,
every tensor the backward reads kept (the student sequence is the query only: tens of tokens).
Backward = explicit, layer by layer, every step a HIP kernel: dense-layer input gradients are the forward
MFMA GEMM on transposed weight copies (built once by ``TrainWeights``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .idefics_engine import IdeficsEngine, IdeficsWeights


def _t(w: torch.Tensor) -> torch.Tensor:
    return w.t().contiguous()


def _unpack_gate_up(packed: torch.Tensor) -> torch.Tensor:
    """Inverse of ops.pack_gate_up: the engine keeps gate|up interleaved in 16-row blocks for the SwiGLU epilogue; the
    backward wants (2I, H) = [gate ; up] (so the student pass can be built from the engine's weights alone)."""
    two_i, K = packed.shape
    blk = packed.view(two_i // 32, 2, 16, K)
    return torch.cat([blk[:, 0].reshape(two_i // 2, K), blk[:, 1].reshape(two_i // 2, K)]).contiguous()


class TrainWeights:
    """Transposed / unfused copies of the language-stack weights needed by the backward (built lazily, once)."""

    def __init__(self, w: IdeficsWeights, sd: Optional[Dict[str, torch.Tensor]] = None):
        """``sd`` (the HF-named state dict) is optional: without it the unfused gate|up copy is recovered from the engine's
        packed buffer (bit-identical rows)."""
        dev = w.device
        self.dec, self.xat = [], []
        for i, D in enumerate(w.dec):
            gu = _unpack_gate_up(D.gu_w)                                                                 # (2I, H) gate | up
            self.dec.append(dict(gu=gu, gu_T=_t(gu), down_T=_t(D.down_w), o_T=_t(D.o_w), qkv_T=_t(D.qkv_w)))
        for j, X in enumerate(w.xat):
            gu = _unpack_gate_up(X.gu_w)
            self.xat.append(dict(gu=gu, gu_T=_t(gu), down_T=_t(X.down_w), o_T=_t(X.o_w), q_T=_t(X.q_w)))
        V = w.lm_head.shape[0]
        head_T = torch.zeros((w.lm_head.shape[1], (V + 7) // 8 * 8), dtype=torch.bfloat16, device=dev)     # (H, V padded to 8)
        head_T[:, :V] = w.lm_head.t()
        self.head_T = head_T
        self.neg_sin = (-w.sin.float()).to(torch.bfloat16).contiguous()


class StudentPass:
    """One hooked forward that keeps what the backward needs, and its backward."""

    def __init__(self, engine: IdeficsEngine, tw: TrainWeights):
        self.e, self.tw = engine, tw

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, attention_mask, pixel_values, image_attention_mask, icv: torch.Tensor,
                hook_layers: Sequence[int], alpha: Optional[torch.Tensor], logits_rows: torch.Tensor,
                image_states: Optional[torch.Tensor] = None):
        """icv (1,n,H) fp32 and alpha (1,n) fp32 as in IdeficsEngine.forward; logits only for `logits_rows`
        (flat b*S+t indices of the answer tokens).  image_states: precomputed perceiver outputs (licv.feature_cache) instead of
        pixel_values.  Returns (logits_rows (R, V) bf16 view, saved state)."""
        e, a, w, tw = self.e, self.e.arch, self.e.w, self.tw
        dev = w.device
        B, S = input_ids.shape
        M, H, nh, hd = B * S, a.hidden_size, a.num_heads, a.head_dim
        if image_states is None:
            with torch.no_grad():
                image_states = e.encode_images(pixel_values)
        Nk, E = image_states.shape[1], image_states.shape[2]
        img_len = a.image_seq_len
        img_mask = image_attention_mask.to(torch.int32).contiguous()
        gate = (img_mask != 0).any(-1).to(torch.float32).reshape(-1).contiguous()
        key_valid = attention_mask.to(torch.int32).contiguous()
        pos = e._position_ids(attention_mask, S).reshape(-1)
        idx_of = {int(l): i for i, l in enumerate(hook_layers)}
        icv = icv.detach().to(device=dev, dtype=torch.float32).contiguous()
        alpha = alpha.detach().to(device=dev, dtype=torch.float32).contiguous() if alpha is not None else None
        img2d = image_states.reshape(B * Nk, E)

        st = dict(B=B, S=S, Nk=Nk, img_len=img_len, img_mask=img_mask, gate=gate, key_valid=key_valid, pos=pos, icv=icv, alpha=alpha,
                  idx_of=idx_of, layers=[], logits_rows=logits_rows)
        h = ops.embed_gather(input_ids.contiguous(), w.embed, w.embed_extra, a.vocab_size).view(M, H)
        for l in range(a.num_layers):
            rec = {}
            if l % a.cross_layer_interval == 0:
                j = l // a.cross_layer_interval
                X, T = w.xat[j], tw.xat[j]
                x = {"h_in": h}
                xn = ops.rmsnorm(h, X.in_ln, a.rms_eps)
                q = ops.linear(xn, X.q_w)
                kv = ops.linear(img2d, X.kv_w)
                x["q_pre"] = q.clone() if X.qn_w is not None else None
                if X.qn_w is not None:
                    ops.rmsnorm(q, X.qn_w, a.rms_eps, out=q, inner=nh, ld_x=H, ld_out=H, rows=M * nh, dim=hd)
                    ops.rmsnorm(kv, X.kn_w, a.rms_eps, out=kv, inner=nh, ld_x=2 * H, ld_out=2 * H, rows=B * Nk * nh, dim=hd)
                x["q"], x["kv"] = q, kv
                o = ops.attention(q, kv, kv.view(-1)[H:], B, S, Nk, nh, nh, hd, S * H, H, Nk * 2 * H, 2 * H, hd ** -0.5, 3,
                                  img_mask=img_mask, img_len=img_len)
                h = ops.linear(o.view(M, H), X.o_w, row_gate=gate, scale=X.gate_attn, residual=h)
                x["h_mid"] = h
                xn = ops.rmsnorm(h, X.post_ln, a.rms_eps)
                gu = ops.linear(xn, T["gu"])
                act = ops.swiglu(gu)
                x["gu"] = gu
                h = ops.linear(act, X.down_w, scale=X.gate_dense, residual=h)
                rec["x"] = x
            D, T = w.dec[l], tw.dec[l]
            rec["h_in"] = h
            xn = ops.rmsnorm(h, D.in_ln, a.rms_eps)
            qkv = ops.linear(xn, D.qkv_w)
            ops.rotary_(qkv, w.cos, w.sin, pos, M, nh, hd, 3 * H, H, 2)
            rec["qkv"] = qkv
            o = ops.attention(qkv, qkv.view(-1)[H:], qkv.view(-1)[2 * H:], B, S, S, nh, nh, hd, S * 3 * H, 3 * H, S * 3 * H, 3 * H,
                              hd ** -0.5, 1, key_valid=key_valid)
            h = ops.linear(o.view(M, H), D.o_w, residual=h)
            rec["h_mid"] = h
            xn = ops.rmsnorm(h, D.post_ln, a.rms_eps)
            gu = ops.linear(xn, T["gu"])
            rec["gu"] = gu
            act = ops.swiglu(gu)
            h = ops.linear(act, D.down_w, residual=h)
            if l in idx_of:
                i = idx_of[l]
                rec["h_pre_hook"] = h
                al = alpha[0, i:i + 1] if alpha is not None else None
                h = ops.inject_renorm(h, icv[0, i], alpha=al)
            st["layers"].append(rec)
        st["h_final"] = h
        xf = ops.rmsnorm(h, w.final_ln, a.rms_eps)
        logits = ops.linear(xf.index_select(0, logits_rows), w.lm_head)
        return logits, st

    # ------------------------------------------------------------------ backward
    def backward(self, st: dict, dlogits_rows: torch.Tensor) -> torch.Tensor:
        """dlogits_rows: (R, V) bf16 grad wrt the logits returned by forward.  Returns d loss / d v_l for every hooked
        layer as (1, n_hooked, H) fp32, where v_l = alpha_l * icv_l is what the hook added."""
        e, a, w, tw = self.e, self.e.arch, self.e.w, self.tw
        B, S, Nk = st["B"], st["S"], st["Nk"]
        M, H, nh, hd, I = B * S, a.hidden_size, a.num_heads, a.head_dim, a.intermediate_size
        dev = w.device
        V = w.lm_head.shape[0]
        n_hooked = len(st["idx_of"])
        grad_v = torch.zeros((1, n_hooked, H), dtype=torch.float32, device=dev)
        # head + final norm
        assert dlogits_rows.shape[1] == tw.head_T.shape[1] and dlogits_rows.is_contiguous(), "pass the padded grad from ops.kl_rows_bwd"
        d_xf_rows = ops.linear(dlogits_rows, tw.head_T)                                              # (R, H) bf16
        d_xf = torch.zeros((M, H), dtype=torch.bfloat16, device=dev)
        d_xf.index_copy_(0, st["logits_rows"], d_xf_rows)
        dh = torch.empty((M, H), dtype=torch.float32, device=dev)
        ops.rmsnorm_bwd(st["h_final"], w.final_ln, d_xf, dh, a.rms_eps, accumulate=False)
        for l in reversed(range(a.num_layers)):
            rec = st["layers"][l]
            D, T = w.dec[l], tw.dec[l]
            if l in st["idx_of"]:
                i = st["idx_of"][l]
                al = st["alpha"][0, i:i + 1] if st["alpha"] is not None else None
                dh, gv = ops.inject_renorm_bwd(rec["h_pre_hook"], st["icv"][0, i], al, dh)
                grad_v[0, i] = gv
            # MLP branch
            d_out = ops.branch_grad(dh)
            d_act = ops.linear(d_out, T["down_T"])
            d_gu = ops.swiglu_bwd(rec["gu"], d_act)
            ops.rmsnorm_bwd_from(rec["h_mid"], D.post_ln, d_gu, T["gu_T"], dh, a.rms_eps, accumulate=True)
            # attention branch
            d_o = ops.branch_grad(dh)
            d_attn = ops.linear(d_o, T["o_T"])
            qkv = rec["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd_small(qkv, qkv.view(-1)[H:], qkv.view(-1)[2 * H:], d_attn, B, S, S, nh, nh, hd, S * 3 * H, 3 * H,
                                    S * 3 * H, 3 * H, hd ** -0.5, 1, dqkv, S * 3 * H, 3 * H, dk=dqkv.view(-1)[H:], dv=dqkv.view(-1)[2 * H:],
                                    dkv_bs=S * 3 * H, dkv_rs=3 * H, key_valid=st["key_valid"])
            ops.rotary_(dqkv, w.cos, tw.neg_sin, st["pos"], M, nh, hd, 3 * H, H, 2)                 # inverse rotation
            ops.rmsnorm_bwd_from(rec["h_in"], D.in_ln, dqkv, T["qkv_T"], dh, a.rms_eps, accumulate=True)
            if "x" in rec:
                x = rec["x"]
                j = l // a.cross_layer_interval
                X, TX = w.xat[j], tw.xat[j]
                d_out = ops.branch_grad(dh, scale=X.gate_dense)
                d_act = ops.linear(d_out, TX["down_T"])
                d_gu = ops.swiglu_bwd(x["gu"], d_act)
                ops.rmsnorm_bwd_from(x["h_mid"], X.post_ln, d_gu, TX["gu_T"], dh, a.rms_eps, accumulate=True)
                d_o = ops.branch_grad(dh, scale=X.gate_attn, row_gate=st["gate"])
                d_attn = ops.linear(d_o, TX["o_T"])
                dq = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
                kv = x["kv"]
                ops.attention_bwd_small(x["q"], kv, kv.view(-1)[H:], d_attn, B, S, Nk, nh, nh, hd, S * H, H, Nk * 2 * H, 2 * H,
                                        hd ** -0.5, 3, dq, S * H, H, img_mask=st["img_mask"], img_len=st["img_len"])
                if X.qn_w is not None:
                    dq_pre = torch.empty_like(dq)
                    ops.rmsnorm_bwd(x["q_pre"], X.qn_w, dq, dq_pre, a.rms_eps, accumulate=False, inner=nh, ld_x=H, ld_dy=H, ld_dx=H,
                                    rows=M * nh, dim=hd)
                    dq = dq_pre
                ops.rmsnorm_bwd_from(x["h_in"], X.in_ln, dq, TX["q_T"], dh, a.rms_eps, accumulate=True)
        return grad_v


# ====================================================================================================== Idefics2
class TrainWeights2:
    """Transposed / unfused copies of the Mistral stack's weights for the backward (Idefics2)."""

    def __init__(self, w, sd: Optional[Dict[str, torch.Tensor]] = None):
        dev = w.device
        self.text = []
        for i, L in enumerate(w.text):
            gu = _unpack_gate_up(L.gu_w)                                                                 # (2I, H) gate | up
            self.text.append(dict(gu=gu, gu_T=_t(gu), down_T=_t(L.down_w), o_T=_t(L.o_w), qkv_T=_t(L.qkv_w)))
        V = w.lm_head.shape[0]
        head_T = torch.zeros((w.lm_head.shape[1], (V + 7) // 8 * 8), dtype=torch.bfloat16, device=dev)
        head_T[:, :V] = w.lm_head.t()
        self.head_T = head_T
        self.neg_sin = (-w.sin.float()).to(torch.bfloat16).contiguous()


class StudentPass2:
    """Idefics2 student pass with gradient: hook on every text layer's MLP BRANCH (ref:config/lmm/idefics2-8B-base.yaml:8).
    Per hooked layer  h_out = h_mid + edit(m):  the stream gradient passes through unchanged and the hook's backward kernel
    turns it into d m (-> MLP backward) and d (alpha*icv)."""

    def __init__(self, engine, tw: TrainWeights2):
        self.e, self.tw = engine, tw

    def forward(self, input_ids, attention_mask, pixel_values, pixel_attention_mask, icv: torch.Tensor,
                hook_layers: Sequence[int], alpha: Optional[torch.Tensor], logits_rows: torch.Tensor):
        e, a, w, tw = self.e, self.e.arch, self.e.w, self.tw
        dev = w.device
        B, S = input_ids.shape
        M, H, nh, nkv, hd = B * S, a.hidden_size, a.num_heads, a.num_kv_heads, a.head_dim
        qd, kd = nh * hd, nkv * hd
        ldq = qd + 2 * kd
        ids = input_ids.to(dev).contiguous()
        with torch.no_grad():
            img = e.encode_images(pixel_values, pixel_attention_mask) if pixel_values is not None else None
        h = ops.embed_gather(ids, w.embed, None, w.embed.shape[0]).view(M, H)
        if img is not None:
            from . import frontend
            frontend.merge_image_rows_(h, ids, img.reshape(-1, H).contiguous(), a.image_token_id)
        key_valid = attention_mask.to(device=dev, dtype=torch.int32).contiguous()
        pos = torch.arange(S, device=dev, dtype=torch.int64).repeat(B).contiguous()
        idx_of = {int(l): i for i, l in enumerate(hook_layers)}
        icv = icv.detach().to(device=dev, dtype=torch.float32).contiguous()
        alpha = alpha.detach().to(device=dev, dtype=torch.float32).contiguous() if alpha is not None else None
        st = dict(B=B, S=S, key_valid=key_valid, pos=pos, icv=icv, alpha=alpha, idx_of=idx_of, layers=[], logits_rows=logits_rows)
        for l, L in enumerate(w.text):
            T = tw.text[l]
            rec = {"h_in": h}
            xn = ops.rmsnorm(h, L.in_ln, a.rms_eps, 1)
            qkv = ops.linear(xn, L.qkv_w)
            ops.rotary_(qkv, w.cos, w.sin, pos, M, nh + nkv, hd, ldq, 0, 1)       # Q heads | K heads contiguous in the fused row
            rec["qkv"] = qkv
            o = ops.attention(qkv, qkv.view(-1)[qd:], qkv.view(-1)[qd + kd:], B, S, S, nh, nkv, hd, S * ldq, ldq, S * ldq, ldq,
                              hd ** -0.5, 1, key_valid=key_valid)
            h = ops.linear(o.view(M, qd), L.o_w, residual=h)
            rec["h_mid"] = h
            xn = ops.rmsnorm(h, L.post_ln, a.rms_eps, 1)
            gu = ops.linear(xn, T["gu"])
            rec["gu"] = gu
            act = ops.swiglu(gu)
            if l in idx_of:
                i = idx_of[l]
                m = ops.linear(act, L.down_w)
                rec["m"] = m
                al = alpha[0, i:i + 1] if alpha is not None else None
                h = ops.inject_renorm_add(m, icv[0, i], h, alpha=al)
            else:
                h = ops.linear(act, L.down_w, residual=h)
            st["layers"].append(rec)
        st["h_final"] = h
        xf = ops.rmsnorm(h, w.final_ln, a.rms_eps, 1)
        logits = ops.linear(xf.index_select(0, logits_rows), w.lm_head)
        return logits, st

    def backward(self, st: dict, dlogits_rows: torch.Tensor) -> torch.Tensor:
        e, a, w, tw = self.e, self.e.arch, self.e.w, self.tw
        B, S = st["B"], st["S"]
        M, H, nh, nkv, hd = B * S, a.hidden_size, a.num_heads, a.num_kv_heads, a.head_dim
        qd, kd, rep = nh * hd, nkv * hd, nh // nkv
        ldq = qd + 2 * kd
        dev = w.device
        grad_v = torch.zeros((1, len(st["idx_of"]), H), dtype=torch.float32, device=dev)
        assert dlogits_rows.shape[1] == tw.head_T.shape[1] and dlogits_rows.is_contiguous(), "pass the padded grad from ops.kl_rows_bwd"
        d_xf = torch.zeros((M, H), dtype=torch.bfloat16, device=dev)
        d_xf.index_copy_(0, st["logits_rows"], ops.linear(dlogits_rows, tw.head_T))
        dh = torch.empty((M, H), dtype=torch.float32, device=dev)
        ops.rmsnorm_bwd(st["h_final"], w.final_ln, d_xf, dh, a.rms_eps, accumulate=False, flavour=1)
        for l in reversed(range(a.num_layers)):
            rec, L, T = st["layers"][l], w.text[l], tw.text[l]
            # MLP branch: hooked -> through the edit's backward, else straight
            if l in st["idx_of"]:
                i = st["idx_of"][l]
                al = st["alpha"][0, i:i + 1] if st["alpha"] is not None else None
                d_m, gv = ops.inject_renorm_bwd(rec["m"], st["icv"][0, i], al, dh)
                grad_v[0, i] = gv
                d_out = ops.branch_grad(d_m)
            else:
                d_out = ops.branch_grad(dh)
            d_act = ops.linear(d_out, T["down_T"])
            d_gu = ops.swiglu_bwd(rec["gu"], d_act)
            ops.rmsnorm_bwd_from(rec["h_mid"], L.post_ln, d_gu, T["gu_T"], dh, a.rms_eps, accumulate=True, flavour=1)
            # attention branch (GQA): dK / dV per query head, then the group sum = backward of repeat_kv
            d_attn = ops.linear(ops.branch_grad(dh), T["o_T"])
            qkv = rec["qkv"]
            dqkv = torch.empty_like(qkv)
            dkv_heads = torch.empty((M, 2 * qd), dtype=torch.bfloat16, device=dev)          # [dK per q head | dV per q head]
            ops.attention_bwd_small(qkv, qkv.view(-1)[qd:], qkv.view(-1)[qd + kd:], d_attn, B, S, S, nh, nkv, hd, S * ldq, ldq,
                                    S * ldq, ldq, hd ** -0.5, 1, dqkv, S * ldq, ldq, dk=dkv_heads, dv=dkv_heads.view(-1)[qd:],
                                    dkv_bs=S * 2 * qd, dkv_rs=2 * qd, key_valid=st["key_valid"])
            ops.head_group_sum(dkv_heads, dqkv.view(-1)[qd:], M, nkv, rep, hd, 2 * qd, ldq)
            ops.head_group_sum(dkv_heads.view(-1)[qd:], dqkv.view(-1)[qd + kd:], M, nkv, rep, hd, 2 * qd, ldq)
            ops.rotary_(dqkv, w.cos, tw.neg_sin, st["pos"], M, nh + nkv, hd, ldq, 0, 1)      # inverse rotation of dQ and dK
            ops.rmsnorm_bwd_from(rec["h_in"], L.in_ln, dqkv, T["qkv_T"], dh, a.rms_eps, accumulate=True, flavour=1)
        return grad_v
