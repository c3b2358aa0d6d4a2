"""Oracle (test infrastructure): hooked greedy / beam-search decoding on the CPU restatement.

Search bookkeeping follows transformers 5.15 ``GenerationMixin._sample`` / ``_beam_search``
(generation/utils.py:3077-3460) driven as ref:inference.py:300-321 does; the model is
``oracle.idefics_ref.forward`` with its KV cache.  Pinned by tests/golden/g5_generate.npz (ids the reference
wrapper + HF generate produced in fp32): ``tests/test_oracle_golden.py::test_g5_generate``.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import idefics_ref as R


@torch.no_grad()
def generate(sd, arch, input_ids, attention_mask, pixel_values, image_attention_mask, icv=None,
             hook_layers: Optional[Sequence[int]] = None, max_new_tokens=5, num_beams=1, length_penalty=1.0,
             min_new_tokens=0, early_stopping=False, eos_token_id=None, pad_token_id=None):
    eos = arch.eos_token_id if eos_token_id is None else eos_token_id
    pad = arch.pad_token_id if pad_token_id is None else pad_token_id
    B, P = input_ids.shape
    nb = num_beams
    max_len = P + max_new_tokens
    hooks = dict(icv=icv, hook_layers=hook_layers) if icv is not None else {}
    image_states = R.image_states_from_pixels(pixel_values, sd, arch)

    # HF expands every input to B*nb rows before the prefill
    ids = input_ids.repeat_interleave(nb, 0)
    am = attention_mask.repeat_interleave(nb, 0)
    iam = image_attention_mask.repeat_interleave(nb, 0)
    img = image_states.repeat_interleave(nb, 0)
    cache = [None] * arch.num_layers

    def step(new_ids, am, iam):
        return R.forward(sd, arch, new_ids, am, image_attention_mask=iam, image_states=img, kv_cache=cache, **hooks)[:, -1, :].float()

    logits = step(ids, am, iam)
    V = logits.shape[-1]
    last_iam = iam[:, -1:, :]

    if nb == 1:
        seq = torch.full((B, max_len), pad, dtype=torch.long)
        seq[:, :P] = input_ids
        unfinished = torch.ones(B, dtype=torch.bool)
        cur = P
        while True:
            nxt = logits.argmax(-1)
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
            seq[:, cur] = nxt
            cur += 1
            if eos is not None:
                unfinished = unfinished & (nxt != eos)
            if cur >= max_len or not bool(unfinished.any()):
                break
            am = torch.cat([am, torch.ones((B, 1), dtype=am.dtype)], 1)
            logits = step(nxt[:, None], am, last_iam)
        return seq[:, :cur]

    keep = 2 * nb
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(keep - nb, dtype=torch.bool)])
    running = torch.full((B, nb, max_len), pad, dtype=torch.long)
    running[:, :, :P] = input_ids[:, None, :]
    finished = running.clone()
    run_scores = torch.zeros((B, nb)); run_scores[:, 1:] = -1e9
    fin_scores = torch.full((B, nb), -1e9)
    is_fin = torch.zeros((B, nb), dtype=torch.bool)
    improve = torch.ones((B, 1), dtype=torch.bool)
    gen_len = torch.zeros((B, nb), dtype=torch.long)
    cur = P

    def gather(t, idx):
        ix = idx
        while ix.dim() < t.dim():
            ix = ix.unsqueeze(-1)
        return torch.gather(t, 1, ix.expand(*idx.shape, *t.shape[2:]))

    while True:
        lp = torch.log_softmax(logits, dim=-1).view(B, nb, V) + run_scores[:, :, None]
        top_lp, top_ix = torch.topk(lp.view(B, nb * V), k=keep)
        src = top_ix // V
        top_seq = gather(running, src)
        top_seq[:, :, cur] = top_ix % V
        hits = (top_seq[:, :, cur] == eos) | (cur + 1 >= max_len)
        run_lp = top_lp + hits.float() * -1.0e9
        nxt_ix = torch.topk(run_lp, k=nb)[1]
        running, run_scores, beam_src = gather(top_seq, nxt_ix), gather(run_lp, nxt_ix), gather(src, nxt_ix)
        just = hits & top_mask[None, :]
        fin_lp = top_lp / ((cur + 1 - P) ** length_penalty)
        fin_lp = fin_lp + (torch.all(is_fin, dim=-1, keepdim=True) & (early_stopping is True)).float() * -1.0e9
        fin_lp = fin_lp + (~improve).float() * -1.0e9 + (~just).float() * -1.0e9
        m_sc = torch.cat([fin_scores, fin_lp], 1)
        best = torch.topk(m_sc, k=nb)[1]
        finished = gather(torch.cat([finished, top_seq], 1), best)
        is_fin = gather(torch.cat([is_fin, just], 1), best)
        gen_len = gather(torch.cat([gen_len, torch.full_like(top_ix, cur + 1 - P)], 1), best)
        fin_scores = gather(m_sc, best)
        flat = (beam_src + torch.arange(B)[:, None] * nb).reshape(-1)
        for i in range(len(cache)):
            cache[i] = (cache[i][0].index_select(0, flat), cache[i][1].index_select(0, flat))
        cur += 1
        best_run = run_scores[:, :1] / (float(cur - P) ** length_penalty)
        worst = torch.where(is_fin, fin_scores.min(dim=1, keepdim=True)[0], torch.full_like(fin_scores, -1.0e9))
        improve = improve & torch.any(best_run > worst, dim=-1, keepdim=True)
        if not (bool(improve.any()) and not (bool(is_fin.all()) and early_stopping is True) and not bool(hits.all())):
            break
        am = torch.cat([am, torch.ones((B * nb, 1), dtype=am.dtype)], 1)
        logits = step(running[:, :, cur - 1].reshape(B * nb, 1), am, last_iam)
    return finished[:, 0, : P + int(gen_len[:, 0].max())]
