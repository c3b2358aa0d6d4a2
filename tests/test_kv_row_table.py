"""Host logic of the beam search's KV cache (licv.idefics_engine.KVCache, CPU tensors): the cache is never moved; a row table names, per
beam row and position, the physical row that holds it.  Reading the history through the table after any sequence of reorders must give
exactly what the round-3 scheme gave by physically gathering the cache at every step (HF's `_reorder_cache`,
transformers/generation/utils.py) — and what `licv_beam_step` writes as the new table (tests/test_beam_step_gpu.py checks the kernel
against `oracle.generate_ref.beam_step`; here: the torch form of the same update)."""
import types

import torch

from licv.idefics_engine import KVCache


def test_row_table_reads_equal_a_physically_reordered_cache():
    g = torch.Generator().manual_seed(0)
    arch = types.SimpleNamespace(num_layers=2, hidden_size=8)
    B, nb, P, new = 3, 3, 5, 4
    cache = KVCache(arch, B, P + new, "cpu", beams=nb)
    width = cache._all.shape[-1]
    cache._all.zero_()
    # prefill: question b's prompt in row b (the first B rows of every layer)
    assert cache.kv[0].shape[0] == B and cache.kv[0].data_ptr() == cache._all[0].data_ptr()
    for l in range(arch.num_layers):
        cache.kv[l][:, :P] = torch.randn(B, P, width, generator=g).to(torch.bfloat16)
    cache.len = P
    # the reference scheme: every beam row owns a full copy of its history, gathered at every reorder
    phys = [cache.kv[l][:, :P].repeat_interleave(nb, 0).clone() for l in range(arch.num_layers)]
    cache.replicate(nb)
    assert cache.rows.shape == (B * nb, P + new) and cache.rows.dtype == torch.int32
    assert torch.equal(cache.rows[:, :P], (torch.arange(B * nb) // nb).to(torch.int32).unsqueeze(1).expand(-1, P))

    def read(l):                                                  # what licv_decode_attn reads for every beam row
        n = cache.len
        pos = torch.arange(n)
        return cache._all[l][cache.rows[:, :n].long(), pos[None, :]]

    for step in range(new):
        for l in range(arch.num_layers):
            assert torch.equal(read(l), phys[l]), f"before step {step}, layer {l}"
        # the decode step appends beam row r's new token to physical row r at position len
        tok = [torch.randn(B * nb, 1, width, generator=g).to(torch.bfloat16) for _ in range(arch.num_layers)]
        assert torch.equal(cache.rows[:, cache.len], torch.arange(B * nb, dtype=torch.int32))
        for l in range(arch.num_layers):
            cache._all[l][:, cache.len] = tok[l][:, 0]
            phys[l] = torch.cat([phys[l], tok[l]], 1)
        cache.len += 1
        if step == new - 1:
            break
        # beam reorder: within every question, beam r continues some source beam (duplicates allowed)
        src = torch.stack([torch.randint(0, nb, (nb,), generator=g) + b * nb for b in range(B)]).reshape(-1)
        cache.reorder(src)
        phys = [p.index_select(0, src) for p in phys]
    for l in range(arch.num_layers):
        assert torch.equal(read(l), phys[l])


def test_a_cache_for_one_beam_per_question_needs_no_table():
    arch = types.SimpleNamespace(num_layers=1, hidden_size=4)
    cache = KVCache(arch, 2, 6, "cpu")
    assert cache.rows is None and cache.kv[0].shape == (2, 6, 8)
