"""Oracle (test infrastructure): hooked greedy / beam-search decoding on the CPU restatement.

Search bookkeeping follows transformers 5.15 ``GenerationMixin._sample`` / ``_beam_search``
(generation/utils.py:3077-3460) driven as ref:inference.py:300-321 does.  Models: ``oracle.idefics_ref.forward`` with
its KV cache (pinned by tests/golden/g5_generate.npz) and ``oracle.idefics2_ref.forward`` re-run over the whole prefix at
every step with HF's generate-time position ids (generation/utils.py:751-773, :975-985; pinned by
tests/golden/g8_generate_idefics2.npz).  Both fixtures hold ids the reference wrapper + HF generate produced in fp32.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import idefics_ref as R
from . import idefics2_ref as R2


class _IdeficsModel:
    def __init__(self, sd, arch, pixel_values, image_attention_mask, nb, hooks):
        self.sd, self.arch, self.hooks = sd, arch, hooks
        self.img = R.image_states_from_pixels(pixel_values, sd, arch).repeat_interleave(nb, 0)
        self.iam = image_attention_mask.repeat_interleave(nb, 0)
        self.cache = [None] * arch.num_layers

    def _fwd(self, ids, am, iam):
        return R.forward(self.sd, self.arch, ids, am, image_attention_mask=iam, image_states=self.img, kv_cache=self.cache,
                         **self.hooks)[:, -1, :].float()

    def prefill(self, ids, am):
        return self._fwd(ids, am, self.iam)

    def step(self, new_ids, am):
        return self._fwd(new_ids, am, self.iam[:, -1:, :])

    def reorder(self, flat):
        for i in range(len(self.cache)):
            self.cache[i] = (self.cache[i][0].index_select(0, flat), self.cache[i][1].index_select(0, flat))


class _Idefics2Model:
    """No cache: the whole prefix is re-run each step (tiny test models).  position ids as HF generate builds them:
    cumsum(attention_mask)-1 with pads at 0 for the prompt, previous+1 for every generated token."""

    def __init__(self, sd, arch, pixel_values, pixel_attention_mask, nb, hooks):
        self.sd, self.arch, self.hooks, self.nb = sd, arch, hooks, nb
        self.img = R2.image_features(pixel_values, pixel_attention_mask, sd, arch)          # (n_real*L, H), rows in batch order
        self.ids = self.pos = None
        self.prompt_len = None

    def _fwd(self, am):
        B = self.ids.shape[0] // self.nb
        P = self.prompt_len
        per_row = (self.ids[:: self.nb, :P] == self.arch.image_token_id).sum(1)            # image rows per original question (prompt only)
        chunks = torch.split(self.img, per_row.tolist())
        img = torch.cat([c for c in chunks for _ in range(self.nb)]) if self.nb > 1 else self.img
        return R2.forward(self.sd, self.arch, self.ids, am, image_hidden_states=img, position_ids=self.pos, scatter_len=P,
                          **self.hooks)[:, -1, :].float()

    def prefill(self, ids, am):
        self.ids = ids
        self.prompt_len = ids.shape[1]
        self.pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 0)
        return self._fwd(am)

    def step(self, new_ids, am):
        self.ids = torch.cat([self.ids, new_ids], 1)
        self.pos = torch.cat([self.pos, self.pos[:, -1:] + 1], 1)
        return self._fwd(am)

    def reorder(self, flat):
        self.ids, self.pos = self.ids.index_select(0, flat), self.pos.index_select(0, flat)


@torch.no_grad()
def generate(sd, arch, input_ids, attention_mask, pixel_values, image_attention_mask, icv=None,
             hook_layers: Optional[Sequence[int]] = None, **kw):
    hooks = dict(icv=icv, hook_layers=hook_layers) if icv is not None else {}
    model = _IdeficsModel(sd, arch, pixel_values, image_attention_mask, kw.get("num_beams", 1), hooks)
    return _decode(model, arch, input_ids, attention_mask, **kw)


@torch.no_grad()
def generate_idefics2(sd, arch, input_ids, attention_mask, pixel_values, pixel_attention_mask, icv=None,
                      hook_layers: Optional[Sequence[int]] = None, **kw):
    hooks = dict(icv=icv, hook_layers=hook_layers) if icv is not None else {}
    model = _Idefics2Model(sd, arch, pixel_values, pixel_attention_mask, kw.get("num_beams", 1), hooks)
    return _decode(model, arch, input_ids, attention_mask, **kw)


def _decode(model, arch, input_ids, attention_mask, max_new_tokens=5, num_beams=1, length_penalty=1.0,
            min_new_tokens=0, early_stopping=False, eos_token_id=None, pad_token_id=None):
    eos = arch.eos_token_id if eos_token_id is None else eos_token_id
    pad = arch.pad_token_id if pad_token_id is None else pad_token_id
    B, P = input_ids.shape
    nb = num_beams
    max_len = P + max_new_tokens
    # HF expands every input to B*nb rows before the prefill
    am = attention_mask.repeat_interleave(nb, 0)
    logits = model.prefill(input_ids.repeat_interleave(nb, 0), am)
    V = logits.shape[-1]

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
            logits = model.step(nxt[:, None], am)
        return seq[:, :cur]

    st = beam_init(input_ids, nb, max_len, eos, pad)
    while True:
        cont, flat = beam_step(st, logits, eos, length_penalty, early_stopping)
        model.reorder(flat)
        if not cont:
            break
        am = torch.cat([am, torch.ones((B * nb, 1), dtype=am.dtype)], 1)
        logits = model.step(st["running"][:, :, st["cur"] - 1].reshape(B * nb, 1), am)
    return st["finished"][:, 0, : P + int(st["gen_len"][:, 0].max())]


def beam_init(input_ids, nb, max_len, eos, pad):
    """State of transformers 5.x `_beam_search` before the first step (generation/utils.py:3290-3340)."""
    B, P = input_ids.shape
    # hf:generation/utils.py:3319 — `output_fill_value = pad_token_id or eos_token_id[0]`: pad id 0 is falsy, EOS fills instead
    fill = pad if (pad or eos is None) else eos
    running = torch.full((B, nb, max_len), fill, dtype=torch.long)
    running[:, :, :P] = input_ids[:, None, :]
    run_scores = torch.zeros((B, nb)); run_scores[:, 1:] = -1e9
    return dict(running=running, finished=running.clone(), run_scores=run_scores, fin_scores=torch.full((B, nb), -1e9),
                is_fin=torch.zeros((B, nb), dtype=torch.bool), improve=torch.ones((B, 1), dtype=torch.bool),
                gen_len=torch.zeros((B, nb), dtype=torch.long), cur=P, P=P, max_len=max_len, nb=nb)


def beam_step(st, logits, eos, length_penalty=1.0, early_stopping=False, suppress_eos=False):
    """One iteration of `_beam_search` (generation/utils.py:3360-3460) on the state `st` (updated in place) given the (B*nb, V) fp32
    logits of the running beams.  Returns (search continues?, flat source-beam index of every new running beam)."""
    B, nb = st["run_scores"].shape
    V = logits.shape[-1]
    cur, P, max_len = st["cur"], st["P"], st["max_len"]
    keep = 2 * nb
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(keep - nb, dtype=torch.bool)])
    running, finished, run_scores, fin_scores = st["running"], st["finished"], st["run_scores"], st["fin_scores"]
    is_fin, improve, gen_len = st["is_fin"], st["improve"], st["gen_len"]

    def gather(t, idx):
        ix = idx
        while ix.dim() < t.dim():
            ix = ix.unsqueeze(-1)
        return torch.gather(t, 1, ix.expand(*idx.shape, *t.shape[2:]))

    lp = torch.log_softmax(logits, dim=-1)
    if suppress_eos:
        lp = lp.clone()
        lp[..., eos] = -float("inf")
    lp = lp.view(B, nb, V) + run_scores[:, :, None]
    top_lp, top_ix = torch.topk(lp.view(B, nb * V), k=keep)
    src = top_ix // V
    top_seq = gather(running, src)
    top_seq[:, :, cur] = top_ix % V
    hits = (top_seq[:, :, cur] == eos) | (cur + 1 >= max_len) if eos is not None else torch.full_like(top_ix, cur + 1 >= max_len, dtype=torch.bool)
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
    cur += 1
    best_run = run_scores[:, :1] / (float(cur - P) ** length_penalty)
    worst = torch.where(is_fin, fin_scores.min(dim=1, keepdim=True)[0], torch.full_like(fin_scores, -1.0e9))
    improve = improve & torch.any(best_run > worst, dim=-1, keepdim=True)
    st.update(running=running, finished=finished, run_scores=run_scores, fin_scores=fin_scores, is_fin=is_fin, improve=improve,
              gen_len=gen_len, cur=cur)
    cont = bool(improve.any()) and not (bool(is_fin.all()) and early_stopping is True) and not bool(hits.all())
    return cont, flat
