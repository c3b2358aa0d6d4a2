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


def _last_logits(engine: IdeficsEngine, **kw) -> torch.Tensor:
    """fp32 logits of the last position of every row."""
    B, S = kw["input_ids"].shape
    rows = torch.arange(B, device=kw["input_ids"].device) * S + (S - 1)
    return engine.forward(**kw, logits_rows=rows).float()


@torch.no_grad()
def generate(engine: IdeficsEngine, input_ids: torch.Tensor, attention_mask: torch.Tensor, pixel_values: torch.Tensor,
             image_attention_mask: torch.Tensor, icv: Optional[torch.Tensor] = None,
             hook_layers: Optional[Sequence[int]] = None, max_new_tokens: int = 5, num_beams: int = 1,
             length_penalty: float = 1.0, min_new_tokens: int = 0, early_stopping=False,
             eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None) -> torch.Tensor:
    a = engine.arch
    dev = input_ids.device
    eos = a.eos_token_id if eos_token_id is None else eos_token_id
    pad = a.pad_token_id if pad_token_id is None else pad_token_id
    B, P = input_ids.shape
    max_len = P + max_new_tokens
    hooks = dict(icv=icv, hook_layers=hook_layers) if icv is not None else {}
    image_states = engine.encode_images(pixel_values)
    nb = num_beams
    cache = KVCache(a, B, max_len, dev)
    logits = _last_logits(engine, input_ids=input_ids, attention_mask=attention_mask, image_states=image_states,
                          image_attention_mask=image_attention_mask, kv_cache=cache, **hooks)
    V = logits.shape[-1]
    last_iam = image_attention_mask[:, -1:, :]

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
            logits = _last_logits(engine, input_ids=nxt[:, None], attention_mask=am, image_states=image_states,
                                  image_attention_mask=last_iam, kv_cache=cache, **hooks)
        return seq[:, :cur]

    # ---- beam search (GenerationMixin._beam_search, transformers 5.x)
    keep = 2 * nb                                                      # max(2, 1 + n_eos) * num_beams
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(keep - nb, dtype=torch.bool)]).to(dev)
    running = torch.full((B, nb, max_len), pad, dtype=torch.long, device=dev)
    running[:, :, :P] = input_ids[:, None, :]
    finished = running.clone()
    run_scores = torch.zeros((B, nb), dtype=torch.float, device=dev)
    run_scores[:, 1:] = -1e9
    fin_scores = torch.full((B, nb), -1e9, dtype=torch.float, device=dev)
    is_fin = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    improve = torch.ones((B, 1), dtype=torch.bool, device=dev)
    gen_len = torch.zeros((B, nb), dtype=torch.long, device=dev)      # generated length of each finished hypothesis

    # replicate the prompt state per beam (HF prefills B*nb identical rows instead)
    cache.kv = [t.repeat_interleave(nb, 0) for t in cache.kv]
    image_states = image_states.repeat_interleave(nb, 0)
    last_iam = last_iam.repeat_interleave(nb, 0)
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
        # reorder the per-beam model state
        flat_src = (beam_src + torch.arange(B, device=dev)[:, None] * nb).reshape(-1)
        cache.reorder(flat_src)
        cur += 1
        # early-stop heuristic (early_stopping=False form): can a running beam still beat the worst finished one?
        best_run = run_scores[:, :1] / (float(cur - P) ** length_penalty)
        worst_fin = torch.where(is_fin, fin_scores.min(dim=1, keepdim=True)[0], torch.full_like(fin_scores, -1.0e9))
        improve = improve & torch.any(best_run > worst_fin, dim=-1, keepdim=True)
        unfinished = bool(improve.any()) and not (bool(is_fin.all()) and early_stopping is True) and not bool(hits.all())
        if not unfinished:
            break
        am = torch.cat([am, torch.ones((B * nb, 1), dtype=am.dtype, device=dev)], 1)
        logits = _last_logits(engine, input_ids=running[:, :, cur - 1].reshape(B * nb, 1), attention_mask=am,
                              image_states=image_states, image_attention_mask=last_iam, kv_cache=cache, **hooks)
    out = finished[:, 0, :]
    out_len = P + int(gen_len[:, 0].max())
    return out[:, :out_len]
