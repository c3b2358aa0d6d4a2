"""Hooked generate token ids against the REFERENCE'S OWN bf16 decode (fixtures g11 / g12: the reference wrapper driving HF
generate with bf16 weights — Idefics as is, Idefics2 under autocast — 16 prompts per padding side, 3-beam search with 5 new
tokens and length_penalty 0 as ref:config/inference.yaml:26-30, greedy, hooks on and off; ref:inference.py:300-321).

Bar: ids are integers, so rows are compared exactly.  A row may only differ from the fixture if the fixture itself marks it
as a near-tie of the REFERENCE: tools/make_golden.py re-ran the reference's decode 24 times with every score moved by -1, 0 or
+1 bf16 ulp at random (two bf16 implementations differ by one ulp on a few logits; bf16 logits tie exactly on several rows) and
stored the fraction of re-runs that reproduced the row.  Rows with stability 1.0 — no comparison of the search within two ulp
— must match bit for bit, and at least 85 % of all rows must match outright.

Fixtures g15 / g16 are the sharp form of the north-star's "token ids bit-exact": the same reference set-up on prompts chosen so
that EVERY row has stability 1.0 (72 jittered re-runs of the reference reproduce all 96 rows per model); there the native
engine must return every id of every row, no floor, no exemption.  g11 / g12 stay as the near-tie study.
"""
import pytest
import torch

from licv.config import IDEFICS2_TINY, IDEFICS_TINY
from licv.synthetic import synth_idefics2_weights, synth_idefics_weights

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"


def _compare(got, z, side, tag, stats):
    gold = T(z[f"{side}_bf16_{tag}_ids"])
    assert got.shape == gold.shape, (got.shape, gold.shape)
    same = (got == gold).all(dim=1)
    sure = T(z[f"{side}_bf16_{tag}_stability"]) >= 1.0
    stats.append((side, tag, int(same.sum()), int(sure.sum()), len(same)))
    assert bool(same[sure].all()), f"{side}/{tag}: rows decided by more than bf16 noise differ: {(~same & sure).nonzero().flatten().tolist()}"
    return same


def _run(w, icv, batch, z, side, stats):
    kw = dict(max_new_tokens=5, length_penalty=0.0, min_new_tokens=0)
    same = [_compare(w.generate(icv=icv, **batch, num_beams=3, **kw).cpu(), z, side, "beam", stats),
            _compare(w.generate(icv=icv, **batch, num_beams=1, **kw).cpu(), z, side, "greedy", stats)]
    w.toggle_intervention(False)
    same.append(_compare(w.generate(icv=icv, **batch, num_beams=1, **kw).cpu(), z, side, "greedy_off", stats))
    w.toggle_intervention(True)
    return torch.cat(same)


@pytest.mark.parametrize("fixture", ["g11_generate_bf16", "g15_generate_bf16_stable"])
def test_idefics_generate_ids_match_reference_bf16_decode(golden, fixture):
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    from lmm_icl_interface import IdeficsInterface
    z = golden(fixture)
    strict = "stable" in fixture
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    sd = synth_idefics_weights(arch, seed=int(z["weights_seed"]) if strict else 121, dtype=torch.float32)
    sd["model.embed_tokens.weight"] *= float(z["embed_scale"])
    sd["lm_head.weight"] *= float(z["head_scale"])
    iface = IdeficsInterface(state_dict=sd, arch=arch, device=DEV)
    w = LearnableICVInterventionLMM(iface, True, -1, "model.model.layers.<LAYER_NUM>", arch.num_layers)
    icv = T(z["icv"]).to(DEV)
    stats, same = [], []
    for side in ("left", "right"):
        batch = {k: T(z[f"{side}_in_{k}"]).to(DEV) for k in ("input_ids", "attention_mask", "pixel_values", "image_attention_mask")}
        same.append(_run(w, icv, batch, z, side, stats))
    print("\n  " + "\n  ".join(f"{s}/{t}: {a}/{n} rows identical ({b} decided by more than bf16 noise)" for s, t, a, b, n in stats))
    frac = float(torch.cat(same).float().mean())
    if strict:
        assert frac == 1.0, f"{fixture}: {int(round((1 - frac) * 96))} of 96 rows differ from the reference's bf16 decode (every row is decided by more than bf16 noise)"
    assert frac >= 0.85, f"only {frac:.2f} of the 96 decoded rows equal the reference's bf16 decode"


@pytest.mark.parametrize("fixture", ["g12_generate_idefics2_bf16", "g16_generate_idefics2_bf16_stable"])
def test_idefics2_generate_ids_match_reference_bf16_autocast_decode(golden, fixture):
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    from lmm_icl_interface import Idefics2Interface
    z = golden(fixture)
    strict = "stable" in fixture
    arch = IDEFICS2_TINY
    sd = synth_idefics2_weights(arch, seed=int(z["weights_seed"]) if strict else 181, dtype=torch.float32)
    sd["model.text_model.embed_tokens.weight"] *= float(z["embed_scale"])
    sd["lm_head.weight"] *= float(z["head_scale"])
    for l in range(arch.num_layers):
        sd[f"model.text_model.layers.{l}.mlp.down_proj.weight"] *= float(z["down_scale"])
    iface = Idefics2Interface(state_dict=sd, arch=arch, device=DEV)
    w = LearnableICVInterventionLMM(iface, True, -1, "model.model.text_model.layers.<LAYER_NUM>.mlp", arch.num_layers)
    icv = T(z["icv"]).to(DEV)
    stats, same = [], []
    for side in ("left", "right"):
        batch = {k: T(z[f"{side}_in_{k}"]).to(DEV) for k in ("input_ids", "attention_mask", "pixel_values", "pixel_attention_mask")}
        same.append(_run(w, icv, batch, z, side, stats))
    print("\n  " + "\n  ".join(f"{s}/{t}: {a}/{n} rows identical ({b} decided by more than bf16 noise)" for s, t, a, b, n in stats))
    frac = float(torch.cat(same).float().mean())
    if strict:
        assert frac == 1.0, f"{fixture}: {int(round((1 - frac) * 96))} of 96 rows differ from the reference's bf16 decode (every row is decided by more than bf16 noise)"
    assert frac >= 0.85, f"only {frac:.2f} of the 96 decoded rows equal the reference's bf16 decode"
