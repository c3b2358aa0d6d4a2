"""licv_beam_step (csrc/beam.hip) against the oracle's restatement of one iteration of transformers' `_beam_search`
(oracle/generate_ref.beam_step, pinned through the generate fixtures g5 / g8 / g11 / g12 / g15 / g16): the same state after every
step of multi-step searches on random logits — running / finished token rows, scores, flags, generated lengths, the KV-cache
reorder indices, the next input tokens and the loop condition — for 1-4 beams, bf16 and fp32 logits, padded row strides, EOS hits,
length penalties, early_stopping and min_new_tokens.  Integers and booleans exactly; scores to fp32 rounding (the kernel evaluates
lp = ((x - max) - log(sum)) + running_score in torch's order, its reduction order over the vocabulary differs)."""
import pytest
import torch

from oracle import generate_ref as G

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(B, nb, V, P, new, dtype, eos, lp, early, min_new, seed, peaked):
    from licv.generation import BeamSearchState
    g = torch.Generator().manual_seed(seed)
    max_len = P + new
    ids = torch.randint(3, V, (B, P), generator=g)
    pad = 0
    st = G.beam_init(ids, nb, max_len, eos, pad)
    fill = pad if (pad or eos is None) else eos
    dev = BeamSearchState(B, nb, P, max_len, fill, ids.to(DEV), eos, lp, early, min_new)
    first = True
    steps = 0
    while True:
        rows = B if first else B * nb
        ld = (V + 7) // 8 * 8 + 8
        buf = torch.randn(rows, ld, generator=g) * (0.5 if peaked else 1.0)
        # the top of every row is decided without ties (bf16 logits tie easily; what torch.topk does with a tie is unspecified, the
        # kernel takes the lower index): 3 * keep tokens per row get distinct values above the noise, 0.5 - 1 apart and off any lattice
        # (values on a lattice made the cumulative scores of different beams collide to within one fp32 ulp)
        n_top = 6 * nb
        for r in range(rows):
            toks = torch.randperm(V, generator=g)[:n_top]
            buf[r, toks] = 6.0 + 0.75 * torch.arange(n_top, dtype=torch.float32)[torch.randperm(n_top, generator=g)] + 0.25 * torch.rand(n_top, generator=g)
            if eos is not None and float(torch.rand((), generator=g)) < (0.5 if peaked else 0.25):
                buf[r, eos] = 6.0 + 0.75 * (n_top - 1) + (0.375 if float(torch.rand((), generator=g)) < 0.5 else -1.875)   # EOS near / at the top
        buf = buf.to(dtype)
        logits = buf[:, :V]                                           # a view with a padded row stride, as the LM head returns it
        ref_logits = logits.float().repeat_interleave(nb, 0) if first else logits.float()
        sup = min_new > 0 and st["cur"] - P < min_new and eos is not None
        cont_ref, flat_ref = G.beam_step(st, ref_logits, eos, lp, early, suppress_eos=sup)
        cont = dev.step(buf.to(DEV)[:, :V], shared_rows=first)
        first = False
        steps += 1
        d = dev.state[0]
        live = st["run_scores"] > -1e8                               # beams whose continuations all stopped hold arbitrary rows (ties at -1e9)
        assert torch.equal(d["run_scores"].cpu() > -1e8, live), f"step {steps}: live running beams"
        assert torch.equal(d["running"].cpu()[live], st["running"][live]), f"step {steps}: running token rows"
        assert torch.equal(d["is_fin"].cpu().bool(), st["is_fin"]), f"step {steps}: is_fin"
        assert torch.equal(d["improve"].cpu().bool().view(-1, 1), st["improve"]), f"step {steps}: improve"
        fin = st["is_fin"]
        assert torch.equal(d["gen_len"].cpu()[fin], st["gen_len"][fin]) and torch.equal(d["finished"].cpu()[fin], st["finished"][fin]), \
            f"step {steps}: finished hypotheses"
        for k, m in (("run_scores", live), ("fin_scores", fin)):
            a_, b_ = d[k].cpu(), st[k]
            if bool(m.any()):
                assert (a_[m] - b_[m]).abs().max() <= 2e-5 * (1 + b_[m].abs().max()), f"step {steps}: {k}"
            assert bool((a_[~m] < -1e8).all()) and bool((b_[~m] < -1e8).all()), f"step {steps}: dead {k}"
        assert cont == cont_ref, f"step {steps}: loop condition"
        assert int(dev.sync.abs().sum()) == 0                        # the counters are left clean for the next step
        if not cont:
            break
        lf = live.reshape(-1)
        assert torch.equal(dev.beam_src_flat.cpu()[lf], flat_ref[lf]), f"step {steps}: KV-cache reorder indices"
        assert torch.equal(dev.next_tokens.cpu()[lf], st["running"][:, :, st["cur"] - 1].reshape(-1)[lf]), f"step {steps}: next tokens"
    out = dev.result().cpu()
    assert torch.equal(out, st["finished"][:, 0, : P + int(st["gen_len"][:, 0].max())])
    return steps


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("nb", [1, 2, 3, 4])
def test_beam_step_matches_the_oracle_step_by_step(nb, dtype):
    total = 0
    for seed, (B, V, P, new, eos, lp, early, min_new, peaked) in enumerate([
            (8, 32002, 7, 5, 2, 0.0, False, 0, False),               # the reference's configuration (ref:config/inference.yaml:26-30)
            (5, 32003, 3, 6, 2, 1.0, False, 0, True),
            (3, 1000, 4, 8, 2, 0.7, True, 0, True),
            (4, 777, 2, 6, None, 1.0, False, 0, False),
            (6, 500, 5, 7, 2, 1.3, False, 3, True)]):
        total += _run(B, nb, V, P, new, dtype, eos, lp, early, min_new, 100 * nb + seed, peaked)
    assert total >= 15


def test_beam_step_rejects_bad_arguments():
    import ctypes as C
    from licv import _lib
    a = _lib.BeamStepArgs()
    assert _lib.lib().licv_beam_step(C.byref(a), None) == -1 and b"null pointer" in _lib.lib().licv_last_error()
