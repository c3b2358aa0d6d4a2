"""The native layer runner (csrc/runner.hip, one C call for the whole Idefics language stack) against the Python layer loop of
licv/idefics_engine.py: same kernels in the same order, so logits must be BIT-IDENTICAL — plain forward with hooks on / off /
on a subset, logits for selected rows, and prefill + single-token decode steps through the KV cache (where the runner also
keeps the step-invariant cross-attention K|V from the prefill instead of re-projecting them)."""
import pytest
import torch

from licv.config import IDEFICS_MID, IDEFICS_TINY
from licv.synthetic import synth_idefics_weights, synth_vqa_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _engines(arch, seed):
    from licv.idefics_engine import IdeficsEngine, IdeficsWeights
    w = IdeficsWeights(synth_idefics_weights(arch, seed=seed, dtype=torch.float32), arch, DEV)
    return IdeficsEngine(w, use_runner=True), IdeficsEngine(w, use_runner=False)


@pytest.mark.parametrize("arch,B,S,N", [(IDEFICS_TINY, 3, 24, 2), (IDEFICS_MID, 2, 96, 5), (IDEFICS_MID, 1, 1, 1)])
def test_runner_forward_is_bit_identical_to_python_loop(arch, B, S, N):
    fast, slow = _engines(arch, 3)
    batch = {k: v.to(DEV) for k, v in synth_vqa_batch(arch, B, max(S, 3 * N + 5), N, seed=4, dtype=torch.bfloat16).items()}
    if S == 1:
        batch = {k: (v[:, :1].contiguous() if k in ("input_ids", "attention_mask", "image_attention_mask") else v) for k, v in batch.items()}
    g = torch.Generator().manual_seed(5)
    icv = (torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.05).to(DEV)
    alpha = torch.full((1, arch.num_layers), 0.3, device=DEV)
    layers = list(range(arch.num_layers))
    for kw in ({}, dict(icv=icv, hook_layers=layers), dict(icv=icv, alpha=alpha, hook_layers=layers),
               dict(icv=icv[:, :2].contiguous(), hook_layers=[1, 3])):
        a, b = fast.forward(**batch, **kw), slow.forward(**batch, **kw)
        assert a.shape == b.shape and torch.equal(a, b), f"hooks {list(kw)}"
    rows = torch.tensor([0, batch["input_ids"].numel() - 1], device=DEV)
    assert torch.equal(fast.forward(**batch, icv=icv, hook_layers=layers, logits_rows=rows),
                       slow.forward(**batch, icv=icv, hook_layers=layers, logits_rows=rows))
    assert fast._runner is not None and slow._runner is None


@pytest.mark.parametrize("side", ["right", "left"])
def test_runner_prefill_and_decode_steps_match_python_loop(side):
    from licv.idefics_engine import KVCache
    arch = IDEFICS_MID
    fast, slow = _engines(arch, 7)
    B, S, N, steps = 3, 20, 2, 4
    batch = synth_vqa_batch(arch, B, S, N, seed=8, min_len=S if side == "right" else 14, dtype=torch.bfloat16, padding_side=side)
    batch = {k: v.to(DEV) for k, v in batch.items()}
    icv = (torch.randn(1, arch.num_layers, arch.hidden_size, generator=torch.Generator().manual_seed(9)) * 0.05).to(DEV)
    layers = list(range(arch.num_layers))
    outs = {}
    for name, eng in (("fast", fast), ("slow", slow)):
        img = eng.encode_images(batch["pixel_values"])
        cache = KVCache(arch, B, S + steps, DEV)
        am, iam = batch["attention_mask"], batch["image_attention_mask"]
        last = torch.arange(B, device=DEV) * S + S - 1
        lg = [eng.forward(batch["input_ids"], am, image_states=img, image_attention_mask=iam, icv=icv, hook_layers=layers, kv_cache=cache,
                          logits_rows=last)]
        iam1 = iam[:, -1:, :].contiguous()
        for t in range(steps):
            nxt = lg[-1].float().argmax(-1)[:, None]
            am = torch.cat([am, torch.ones((B, 1), dtype=am.dtype, device=DEV)], 1)
            lg.append(eng.forward(nxt, am, image_states=img, image_attention_mask=iam1, icv=icv, hook_layers=layers, kv_cache=cache,
                                  logits_rows=torch.arange(B, device=DEV)))
        outs[name] = lg
        assert cache.len == S + steps
        assert (cache.xkv is not None) == (name == "fast")            # only the runner keeps the cross-attention K|V
    for t, (a, b) in enumerate(zip(outs["fast"], outs["slow"])):
        assert torch.equal(a, b), f"step {t}"
