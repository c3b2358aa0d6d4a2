"""Hooked greedy / beam-search decoding on the native engine (ref:inference.py:300-321 `generate_answers`;
ref:config/inference.yaml:26-30: max_new_tokens 5, num_beams 3, length_penalty 0.0, min_new_tokens 0).

The search bookkeeping restates transformers' ``GenerationMixin._sample`` / ``_beam_search`` (5.x vectorised
form: top-2K continuations, running vs finished beam sets, the ``early_stopping=False`` heuristic) so that
token ids are bit-identical for identical logits.  The model side is the native engine: one prefill on B rows
(the KV cache is then replicated per beam instead of prefilling B*beams identical rows), then single-token
steps on (B*beams, 1) with the ICV hook firing at every step, as in the reference.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .idefics_engine import IdeficsEngine, KVCache


def _last_rows(B: int, S: int, dev) -> torch.Tensor:
    return torch.arange(B, device=dev) * S + (S - 1)


class _IdeficsDecoder:
    """Model side of the search for Idefics: image states + per-token image mask + KV cache."""

    def __init__(self, engine: IdeficsEngine, pixel_values, image_attention_mask, batch, max_len, hooks):
        self.e, self.hooks = engine, hooks
        self.image_states = engine.encode_images(pixel_values)
        self.iam = image_attention_mask
        self.cache = KVCache(engine.arch, batch, max_len, engine.w.device)

    def _fwd(self, ids, am, iam):
        B, S = ids.shape
        return self.e.forward(input_ids=ids, attention_mask=am, image_states=self.image_states, image_attention_mask=iam,
                              kv_cache=self.cache, logits_rows=_last_rows(B, S, ids.device), **self.hooks).float()

    def prefill(self, ids, am):
        out = self._fwd(ids, am, self.iam)
        self.iam = self.iam[:, -1:, :]
        return out

    def step(self, new_ids, am):
        return self._fwd(new_ids, am, self.iam)

    def replicate(self, nb):                       # HF prefills B*nb identical rows; replicate the prompt state instead
        self.cache.replicate(nb)
        if self.cache.xkv is not None:             # beams of one question share the image side: replicated once, never reordered
            self.cache.xkv = [t.repeat_interleave(nb, 0) for t in self.cache.xkv]
        self.image_states = self.image_states.repeat_interleave(nb, 0)
        self.iam = self.iam.repeat_interleave(nb, 0)

    def reorder(self, flat):
        self.cache.reorder(flat)


class _Idefics2Decoder:
    """Idefics2: image hidden states only feed the prefill (they are scattered into the prompt embeddings); decode steps
    need the KV cache and HF's generate-time position ids (cumsum(mask)-1 with pads at 0, then previous+1 per step:
    transformers generation/utils.py:751-773, :975-985)."""

    def __init__(self, engine, pixel_values, pixel_attention_mask, batch, max_len, hooks):
        from .idefics2_engine import KVCache2
        self.e, self.hooks = engine, hooks
        self.img = engine.encode_images(pixel_values, pixel_attention_mask) if pixel_values is not None else None
        self.cache = KVCache2(engine.arch, batch, max_len, engine.w.device)
        self.pos = None

    def _fwd(self, ids, am, img):
        B, S = ids.shape
        return self.e.forward(input_ids=ids, attention_mask=am, image_hidden_states=img, position_ids=self.pos, kv_cache=self.cache,
                              logits_rows=_last_rows(B, S, ids.device), **self.hooks).float()

    def prefill(self, ids, am):
        self.pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 0)
        out = self._fwd(ids, am, self.img)
        self.pos = self.pos[:, -1:]
        return out

    def step(self, new_ids, am):
        self.pos = self.pos + 1
        return self._fwd(new_ids, am, None)

    def replicate(self, nb):
        self.cache.replicate(nb)
        self.pos = self.pos.repeat_interleave(nb, 0)

    def reorder(self, flat):
        self.cache.reorder(flat)
        self.pos = self.pos.index_select(0, flat)


@torch.no_grad()
def generate(engine: IdeficsEngine, input_ids: torch.Tensor, attention_mask: torch.Tensor, pixel_values: torch.Tensor,
             image_attention_mask: torch.Tensor, icv: Optional[torch.Tensor] = None,
             hook_layers: Optional[Sequence[int]] = None, max_new_tokens: int = 5, num_beams: int = 1, **kw) -> torch.Tensor:
    hooks = dict(icv=icv, hook_layers=hook_layers) if icv is not None else {}
    model = _IdeficsDecoder(engine, pixel_values, image_attention_mask, input_ids.shape[0], input_ids.shape[1] + max_new_tokens, hooks)
    return _decode(model, engine.arch, input_ids, attention_mask, max_new_tokens=max_new_tokens, num_beams=num_beams, **kw)


@torch.no_grad()
def generate_idefics2(engine, input_ids: torch.Tensor, attention_mask: torch.Tensor, pixel_values: Optional[torch.Tensor] = None,
                      pixel_attention_mask: Optional[torch.Tensor] = None, icv: Optional[torch.Tensor] = None,
                      hook_layers: Optional[Sequence[int]] = None, max_new_tokens: int = 5, num_beams: int = 1, **kw) -> torch.Tensor:
    hooks = dict(icv=icv, hook_layers=hook_layers) if icv is not None else {}
    model = _Idefics2Decoder(engine, pixel_values, pixel_attention_mask, input_ids.shape[0], input_ids.shape[1] + max_new_tokens, hooks)
    return _decode(model, engine.arch, input_ids, attention_mask, max_new_tokens=max_new_tokens, num_beams=num_beams, **kw)


def _decode(model, a, input_ids: torch.Tensor, attention_mask: torch.Tensor, max_new_tokens: int = 5, num_beams: int = 1,
            length_penalty: float = 1.0, min_new_tokens: int = 0, early_stopping=False,
            eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None) -> torch.Tensor:
    dev = input_ids.device
    eos = a.eos_token_id if eos_token_id is None else eos_token_id
    pad = a.pad_token_id if pad_token_id is None else pad_token_id
    B, P = input_ids.shape
    max_len = P + max_new_tokens
    nb = num_beams
    logits = model.prefill(input_ids, attention_mask)
    V = logits.shape[-1]

    def suppress_eos(lp, n_generated):
        if min_new_tokens > 0 and n_generated < min_new_tokens and eos is not None:
            lp = lp.clone()
            lp[..., eos] = -float("inf")
        return lp

    if nb == 1:                                                       # ---- greedy (GenerationMixin._sample)
        seq = torch.full((B, max_len), pad, dtype=torch.long, device=dev)
        seq[:, :P] = input_ids
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        am = attention_mask
        cur = P
        while True:
            scores = suppress_eos(logits, cur - P)
            nxt = scores.argmax(-1)
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
            seq[:, cur] = nxt
            cur += 1
            if eos is not None:
                unfinished = unfinished & (nxt != eos)
            if cur >= max_len or not bool(unfinished.any()):
                break
            am = torch.cat([am, torch.ones((B, 1), dtype=am.dtype, device=dev)], 1)
            logits = model.step(nxt[:, None], am)
        return seq[:, :cur]

    # ---- beam search (GenerationMixin._beam_search, transformers 5.x)
    keep = 2 * nb                                                      # max(2, 1 + n_eos) * num_beams
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(keep - nb, dtype=torch.bool)]).to(dev)
    # hf:generation/utils.py:3319 — `output_fill_value = pad_token_id or eos_token_id[0]`: a pad id of 0 (Idefics' <unk>) is falsy
    # there, so finished beams are padded with EOS, not with the pad id (greedy above does use the pad id)
    fill = pad if (pad or eos is None) else eos
    running = torch.full((B, nb, max_len), fill, dtype=torch.long, device=dev)
    running[:, :, :P] = input_ids[:, None, :]
    finished = running.clone()
    run_scores = torch.zeros((B, nb), dtype=torch.float, device=dev)
    run_scores[:, 1:] = -1e9
    fin_scores = torch.full((B, nb), -1e9, dtype=torch.float, device=dev)
    is_fin = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    improve = torch.ones((B, 1), dtype=torch.bool, device=dev)
    gen_len = torch.zeros((B, nb), dtype=torch.long, device=dev)      # generated length of each finished hypothesis

    # replicate the prompt state per beam (HF prefills B*nb identical rows instead)
    model.replicate(nb)
    am = attention_mask.repeat_interleave(nb, 0)
    logits = logits.repeat_interleave(nb, 0)
    cur = P

    def gather(t, idx):                                                # (B, n, ...) gathered along dim 1
        ix = idx
        while ix.dim() < t.dim():
            ix = ix.unsqueeze(-1)
        return torch.gather(t, 1, ix.expand(*idx.shape, *t.shape[2:]))

    while True:
        lp = torch.log_softmax(logits, dim=-1)
        lp = suppress_eos(lp, cur - P).view(B, nb, V) + run_scores[:, :, None]
        top_lp, top_ix = torch.topk(lp.view(B, nb * V), k=keep)
        src_beam = top_ix // V
        top_seq = gather(running, src_beam)
        top_seq[:, :, cur] = top_ix % V
        hits = (cur + 1 >= max_len) | ((top_seq[:, :, cur] == eos) if eos is not None else torch.zeros_like(top_ix, dtype=torch.bool))
        # next running beams: best `nb` continuations that did not just stop
        run_lp = top_lp + hits.float() * -1.0e9
        nxt_ix = torch.topk(run_lp, k=nb)[1]
        running = gather(top_seq, nxt_ix)
        run_scores = gather(run_lp, nxt_ix)
        beam_src = gather(src_beam, nxt_ix)
        # finished set: only the top `nb` candidates may finalise
        just = hits & top_mask[None, :]
        fin_lp = top_lp / ((cur + 1 - P) ** length_penalty)
        fin_lp = fin_lp + (torch.all(is_fin, dim=-1, keepdim=True) & (early_stopping is True)).float() * -1.0e9
        fin_lp = fin_lp + (~improve).float() * -1.0e9
        fin_lp = fin_lp + (~just).float() * -1.0e9
        m_seq = torch.cat([finished, top_seq], 1)
        m_sc = torch.cat([fin_scores, fin_lp], 1)
        m_fin = torch.cat([is_fin, just], 1)
        m_len = torch.cat([gen_len, torch.full_like(top_ix, cur + 1 - P)], 1)
        best = torch.topk(m_sc, k=nb)[1]
        finished, fin_scores, is_fin, gen_len = gather(m_seq, best), gather(m_sc, best), gather(m_fin, best), gather(m_len, best)
        cur += 1
        # early-stop heuristic (early_stopping=False form): can a running beam still beat the worst finished one?
        best_run = run_scores[:, :1] / (float(cur - P) ** length_penalty)
        worst_fin = torch.where(is_fin, fin_scores.min(dim=1, keepdim=True)[0], torch.full_like(fin_scores, -1.0e9))
        improve = improve & torch.any(best_run > worst_fin, dim=-1, keepdim=True)
        # one device -> host read per step (the loop's only synchronisation point)
        unfinished_t = improve.any() & ~hits.all()
        if early_stopping is True:
            unfinished_t = unfinished_t & ~is_fin.all()
        if not bool(unfinished_t):
            break
        # reorder the per-beam model state (after the exit test: the last step's reorder would feed no further forward)
        flat_src = (beam_src + torch.arange(B, device=dev)[:, None] * nb).reshape(-1)
        model.reorder(flat_src)
        am = torch.cat([am, torch.ones((B * nb, 1), dtype=am.dtype, device=dev)], 1)
        logits = model.step(running[:, :, cur - 1].reshape(B * nb, 1), am)
    out = finished[:, 0, :]
    out_len = P + int(gen_len[:, 0].max())
    return out[:, :out_len]
