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

    def __init__(self, engine: IdeficsEngine, pixel_values, image_attention_mask, batch, max_len, hooks, beams: int = 1):
        self.e, self.hooks = engine, hooks
        self.image_states = engine.encode_images(pixel_values)
        self.iam = image_attention_mask
        self.cache = KVCache(engine.arch, batch, max_len, engine.w.device, beams=beams)

    def _fwd(self, ids, am, iam):
        B, S = ids.shape
        return self.e.forward(input_ids=ids, attention_mask=am, image_states=self.image_states, image_attention_mask=iam,
                              kv_cache=self.cache, logits_rows=_last_rows(B, S, ids.device), **self.hooks)

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

    def kv_rows(self):                             # the cache's row table (after replicate): licv_beam_step keeps it up to date itself
        return self.cache.rows

    def reorder_with_rows(self, flat, rows):
        self.cache.set_rows(rows)


class _Idefics2Decoder:
    """Idefics2: image hidden states only feed the prefill (they are scattered into the prompt embeddings); decode steps
    need the KV cache and HF's generate-time position ids (cumsum(mask)-1 with pads at 0, then previous+1 per step:
    transformers generation/utils.py:751-773, :975-985)."""

    def __init__(self, engine, pixel_values, pixel_attention_mask, batch, max_len, hooks, beams: int = 1):
        from .idefics2_engine import KVCache2
        self.e, self.hooks = engine, hooks
        self.img = engine.encode_images(pixel_values, pixel_attention_mask) if pixel_values is not None else None
        self.cache = KVCache2(engine.arch, batch, max_len, engine.w.device, beams=beams)
        self.pos = None

    def _fwd(self, ids, am, img):
        B, S = ids.shape
        return self.e.forward(input_ids=ids, attention_mask=am, image_hidden_states=img, position_ids=self.pos, kv_cache=self.cache,
                              logits_rows=_last_rows(B, S, ids.device), **self.hooks)

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

    def kv_rows(self):
        return self.cache.rows

    def reorder_with_rows(self, flat, rows):
        self.cache.set_rows(rows)
        self.pos = self.pos.index_select(0, flat)


@torch.no_grad()
def generate(engine: IdeficsEngine, input_ids: torch.Tensor, attention_mask: torch.Tensor, pixel_values: torch.Tensor,
             image_attention_mask: torch.Tensor, icv: Optional[torch.Tensor] = None,
             hook_layers: Optional[Sequence[int]] = None, max_new_tokens: int = 5, num_beams: int = 1, **kw) -> torch.Tensor:
    hooks = dict(icv=icv, hook_layers=hook_layers) if icv is not None else {}
    model = _IdeficsDecoder(engine, pixel_values, image_attention_mask, input_ids.shape[0], input_ids.shape[1] + max_new_tokens, hooks, beams=num_beams)
    return _decode(model, engine.arch, input_ids, attention_mask, max_new_tokens=max_new_tokens, num_beams=num_beams, **kw)


@torch.no_grad()
def generate_idefics2(engine, input_ids: torch.Tensor, attention_mask: torch.Tensor, pixel_values: Optional[torch.Tensor] = None,
                      pixel_attention_mask: Optional[torch.Tensor] = None, icv: Optional[torch.Tensor] = None,
                      hook_layers: Optional[Sequence[int]] = None, max_new_tokens: int = 5, num_beams: int = 1, **kw) -> torch.Tensor:
    hooks = dict(icv=icv, hook_layers=hook_layers) if icv is not None else {}
    model = _Idefics2Decoder(engine, pixel_values, pixel_attention_mask, input_ids.shape[0], input_ids.shape[1] + max_new_tokens, hooks, beams=num_beams)
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
    logits = model.prefill(input_ids, attention_mask)          # (B, V) rows in the model's dtype (a view with a padded row stride)

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
            scores = suppress_eos(logits.float(), cur - P)
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

    # ---- beam search (GenerationMixin._beam_search, transformers 5.x): ONE kernel per step does the log-softmax, the top 2K
    # continuations per question and the whole bookkeeping (csrc/beam.hip; the torch restatement it replaced was ~60 launches a step)
    # hf:generation/utils.py:3319 — `output_fill_value = pad_token_id or eos_token_id[0]`: a pad id of 0 (Idefics' <unk>) is falsy
    # there, so finished beams are padded with EOS, not with the pad id (greedy above does use the pad id)
    fill = pad if (pad or eos is None) else eos
    # the prompt state is shared by a question's beams (HF prefills B*nb identical rows instead): no copy of the KV cache, only its row
    # table, which the search kernel then keeps up to date itself; the first step reads the B prefill rows directly
    model.replicate(nb)
    table = model.kv_rows() if hasattr(model, "kv_rows") and hasattr(model, "reorder_with_rows") else None
    search = BeamSearchState(B, nb, P, max_len, fill, input_ids, eos, length_penalty, early_stopping, min_new_tokens, kv_rows=table)
    am = attention_mask.repeat_interleave(nb, 0)
    first = True
    while True:
        unfinished = search.step(logits, shared_rows=first)            # one device -> host read per step (the loop's only sync point)
        first = False
        if not unfinished:
            break
        # reorder the per-beam model state (after the exit test: the last step's reorder would feed no further forward)
        if search.kv_rows is not None:
            model.reorder_with_rows(search.beam_src_flat, search.kv_rows)
        else:
            model.reorder(search.beam_src_flat)
        am = torch.cat([am, torch.ones((B * nb, 1), dtype=am.dtype, device=dev)], 1)
        logits = model.step(search.next_tokens.view(B * nb, 1), am)
    return search.result()


class BeamSearchState:
    """Device-side state of one beam search (running / finished token rows, scores, flags), advanced by `licv_beam_step`.
    Two copies of every buffer: the kernel gathers rows of the old state into the new one."""

    def __init__(self, B, nb, P, max_len, fill, input_ids, eos, length_penalty, early_stopping, min_new_tokens, kv_rows=None):
        dev = input_ids.device
        self.B, self.nb, self.P, self.max_len, self.cur = B, nb, P, max_len, P
        self.eos = -1 if eos is None else int(eos)
        self.length_penalty, self.early_stopping, self.min_new_tokens = float(length_penalty), early_stopping is True, int(min_new_tokens)
        running = torch.full((B, nb, max_len), fill, dtype=torch.long, device=dev)
        running[:, :, :P] = input_ids[:, None, :]
        run_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
        run_scores[:, 1:] = -1e9
        st = dict(running=running, finished=running.clone(), run_scores=run_scores,
                  fin_scores=torch.full((B, nb), -1e9, dtype=torch.float32, device=dev),
                  is_fin=torch.zeros((B, nb), dtype=torch.uint8, device=dev), improve=torch.ones((B,), dtype=torch.uint8, device=dev),
                  gen_len=torch.zeros((B, nb), dtype=torch.long, device=dev))
        self.state = [st, {k: torch.empty_like(v) for k, v in st.items()}]
        self.beam_src_flat = torch.empty((B * nb,), dtype=torch.long, device=dev)
        self.next_tokens = torch.empty((B * nb,), dtype=torch.long, device=dev)
        self.flags = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.sync = torch.zeros((4,), dtype=torch.int32, device=dev)
        from . import _lib
        self.scratch = torch.empty((int(_lib.lib().licv_beam_step_scratch_bytes(B, nb)),), dtype=torch.uint8, device=dev)
        # optional: the KV cache's row table (B*nb, cache max_len) int32, ping-ponged like the rest of the state
        self.kv_rows = kv_rows
        self._kv_rows_next = torch.empty_like(kv_rows) if kv_rows is not None else None

    def step(self, logits: torch.Tensor, shared_rows: bool = False) -> bool:
        """logits: (B*nb, V) rows (bf16 or fp32, row stride = stride(0)); shared_rows: (B, V) rows of the prefill that all beams of a
        question share.  Returns whether the search continues (reads one int32 back)."""
        import ctypes as C
        from . import _lib
        from .ops import _dt, _p, _stream, check
        assert logits.dim() == 2 and logits.stride(1) == 1 and logits.shape[0] == (self.B if shared_rows else self.B * self.nb)
        a = _lib.BeamStepArgs()
        a.logits, a.logits_dtype, a.ld = logits.data_ptr(), _dt(logits), logits.stride(0)
        a.q_stride_rows, a.beam_stride_rows = (1, 0) if shared_rows else (self.nb, 1)
        a.B, a.nb, a.V, a.max_len, a.cur, a.P = self.B, self.nb, logits.shape[1], self.max_len, self.cur, self.P
        a.eos = self.eos
        a.suppress_eos = 1 if (self.min_new_tokens > 0 and self.cur - self.P < self.min_new_tokens and self.eos >= 0) else 0
        a.length_penalty, a.early_stopping = self.length_penalty, 1 if self.early_stopping else 0
        src, dst = self.state
        for k in ("running", "finished", "run_scores", "fin_scores", "is_fin", "improve", "gen_len"):
            setattr(a, k + "_in", src[k].data_ptr())
            setattr(a, k + "_out", dst[k].data_ptr())
        a.beam_src_flat, a.next_tokens = self.beam_src_flat.data_ptr(), self.next_tokens.data_ptr()
        a.flags, a.sync = self.flags.data_ptr(), self.sync.data_ptr()
        a.scratch, a.scratch_bytes = self.scratch.data_ptr(), self.scratch.numel()
        if self.kv_rows is not None:
            assert self.kv_rows.dtype == torch.int32 and self.kv_rows.is_contiguous() and self.kv_rows.shape[0] == self.B * self.nb
            a.kv_rows_in, a.kv_rows_out, a.kv_ld = self.kv_rows.data_ptr(), self._kv_rows_next.data_ptr(), self.kv_rows.shape[1]
        check(_lib.lib().licv_beam_step(C.byref(a), _stream(logits)))
        if self.kv_rows is not None:
            self.kv_rows, self._kv_rows_next = self._kv_rows_next, self.kv_rows
        self.state = [dst, src]
        self.cur += 1
        return bool(int(self.flags[0]))

    def result(self) -> torch.Tensor:
        st = self.state[0]
        out_len = self.P + int(st["gen_len"][:, 0].max())
        return st["finished"][:, 0, :out_len]
