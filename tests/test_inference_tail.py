"""`generate_answers` / `icv_inference` of the inference mirror (ref:inference.py:246-321) with a fake tokenizer and processor:
ICV scaling, prompt strip at attention_mask.shape[1], batch_decode(skip_special_tokens=True), the bs-wide last chunk.
CPU half: a stub model, so the tail is checked without a GPU; GPU half: the native interface's hooked generate end to end."""
import types

import pytest
import torch

import inference as I


class FakeTokenizer:
    """ids -> "t<id>" words; special ids (pad 0, bos 1, eos 2) are dropped when skip_special_tokens is set."""
    pad_token_id, bos_token_id, eos_token_id, padding_side = 0, 1, 2, "left"
    special = {0, 1, 2}

    def batch_decode(self, rows, skip_special_tokens=False):
        assert all(isinstance(r, list) for r in rows)
        return [" ".join(f"t{i}" for i in r if not (skip_special_tokens and i in self.special)) for r in rows]


class StubModel:
    """Returns prompt + fixed continuation and records what it was called with."""
    def __init__(self, cont):
        self.cont, self.calls = cont, []
        self.lmm = types.SimpleNamespace(device=torch.device("cpu"))

    def generate(self, icv=None, **kw):
        self.calls.append(dict(icv=icv, **kw))
        ids = kw["input_ids"]
        return torch.cat([ids, self.cont[: ids.shape[0]]], dim=1)


def test_generate_answers_scales_icv_strips_prompt_and_decodes():
    tok = FakeTokenizer()
    proc = types.SimpleNamespace(tokenizer=tok)
    ids = torch.tensor([[0, 0, 1, 7, 8], [1, 5, 6, 7, 9]])
    am = (ids != 0).long()
    cont = torch.tensor([[11, 12, 2, 2, 2], [13, 2, 2, 2, 2]])
    model = StubModel(cont)
    vec = torch.arange(6, dtype=torch.float32).reshape(1, 2, 3)
    alpha = torch.tensor([[0.5, 2.0]])
    out = I.generate_answers(dict(input_ids=ids, attention_mask=am), model, proc, dict(num_beams=3, max_new_tokens=5),
                             in_context_vector=vec, alpha=alpha)
    assert out == ["t11 t12", "t13"]                                  # prompt (incl. its pads) gone, eos/pad skipped
    call = model.calls[0]
    assert torch.equal(call["icv"], alpha.unsqueeze(-1) * vec) and call["num_beams"] == 3 and call["max_new_tokens"] == 5
    # no vector: the hooks get icv=None (the reference's ICL / zero-shot baseline call)
    I.generate_answers(dict(input_ids=ids, attention_mask=am), model, proc, {})
    assert model.calls[1]["icv"] is None


def test_icv_inference_batches_and_keeps_short_last_chunk_bs_wide():
    tok = FakeTokenizer()
    seen = []

    def prepare_input(prompts):
        seen.append(prompts)
        n = len(prompts)
        return dict(input_ids=torch.full((n, 3), 4), attention_mask=torch.ones(n, 3, dtype=torch.long))

    proc = types.SimpleNamespace(tokenizer=tok, prepare_input=prepare_input)
    pm = types.SimpleNamespace(gen_query_text_without_label=lambda s: f"Question:{s['question']} Short answer:")
    ds = [dict(image=f"img{i}", question=f"q{i}", question_id=i) for i in range(5)]
    cont = torch.tensor([[20 + i, 2] for i in range(2)])
    model = StubModel(cont)
    res = I.icv_inference(ds, model, pm, proc, bs=2, generate_kwargs={}, instruction="Answer briefly.")
    assert list(res) == [0, 1, 2, 3, 4]
    assert res[3] == {"prediction": "t21", "question": "q3", "question_id": 3}           # image dropped, order kept
    assert [len(p) for p in seen] == [2, 2, 2]                                           # last chunk still bs prompts wide
    assert seen[2][1] == ["Answer briefly."]                                             # ... the spare one instruction-only
    assert seen[0][0] == ["Answer briefly.", "img0", "Question:q0 Short answer:"]
    assert "image" in ds[0]                                                              # caller's samples are not mutated


@pytest.mark.gpu
def test_generate_answers_on_the_native_interface(golden):
    """End to end on the GPU: LearnableICVInterventionLMM over IdeficsInterface, fixture g5's prompts; the decoded strings are
    exactly the tail of the ids the wrapper's own generate returns for alpha * vector."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    from licv.config import IDEFICS_TINY
    from licv.synthetic import synth_idefics_weights
    from lmm_icl_interface import IdeficsInterface
    T = torch.from_numpy
    z = golden("g5_generate")
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    sd = synth_idefics_weights(arch, seed=21, dtype=torch.float32)
    sd["model.embed_tokens.weight"] *= float(z["embed_scale"]); sd["lm_head.weight"] *= float(z["head_scale"])
    iface = IdeficsInterface(state_dict=sd, arch=arch, device="cuda")
    w = LearnableICVInterventionLMM(iface, True, -1, "model.model.layers.<LAYER_NUM>", arch.num_layers)
    tok = FakeTokenizer()
    tok.special = {arch.pad_token_id, arch.bos_token_id, arch.eos_token_id}
    proc = types.SimpleNamespace(tokenizer=tok)
    batch = {k: T(z[f"left_in_{k}"]).to("cuda") for k in ("input_ids", "attention_mask", "pixel_values", "image_attention_mask")}
    vec = T(z["icv"]).to("cuda")
    alpha = torch.full((1, arch.num_layers), 0.5, device="cuda")
    kw = dict(max_new_tokens=5, num_beams=3, length_penalty=0.0, min_new_tokens=0)
    texts = I.generate_answers(batch, w, proc, kw, in_context_vector=vec, alpha=alpha)
    ids = w.generate(icv=alpha.unsqueeze(-1) * vec, **batch, **kw).cpu()
    S = batch["attention_mask"].shape[1]
    assert ids.shape[1] > S and len(texts) == ids.shape[0]
    assert texts == tok.batch_decode([r[S:] for r in ids.tolist()], skip_special_tokens=True)
    assert any(t for t in texts)
