"""smoke(): one tiny hooked Idefics forward on cuda:0 through liblicv_hip.so, checked against the CPU oracle."""
from __future__ import annotations

import torch


def run_smoke(verbose: bool = True) -> float:
    from licv import _lib
    from licv.config import IDEFICS_MID
    from licv.idefics_engine import IdeficsEngine, IdeficsWeights
    from licv.synthetic import synth_idefics_weights, synth_vqa_batch
    from oracle import idefics_ref as R              # the checker (test infrastructure), never the product path

    _lib.check_exports()
    arch = IDEFICS_MID
    sd32 = synth_idefics_weights(arch, seed=3, dtype=torch.float32)
    eng = IdeficsEngine(IdeficsWeights(sd32, arch, "cuda:0"))
    batch = synth_vqa_batch(arch, 2, 40, 2, seed=4, min_len=33, dtype=torch.float32)
    layers = list(range(arch.num_layers))
    icv = torch.randn(1, len(layers), arch.hidden_size, generator=torch.Generator().manual_seed(5)) * 0.05
    out = eng.forward(**{k: v.to("cuda:0") for k, v in batch.items()}, icv=icv.to("cuda:0"), hook_layers=layers)
    torch.cuda.synchronize()
    ref = {}
    for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        sd = {k: v.to(dt) for k, v in sd32.items()}
        kw = dict(batch)
        kw["pixel_values"] = batch["pixel_values"].to(dt)
        with torch.no_grad():
            ref[name] = R.forward(sd, arch, **kw, icv=icv, hook_layers=layers).float()
    got = out.float().cpu()
    scale = float(ref["f32"].abs().max())
    err = float((got - ref["bf16"]).abs().max())
    spread = float((ref["bf16"] - ref["f32"]).abs().max())
    assert torch.isfinite(got).all(), "non-finite logits"
    assert err <= 1.5e-2 * scale, f"smoke parity: |hip - oracle| = {err:.3e} at scale {scale:.3e}"
    assert float((got - ref["f32"]).abs().max()) <= 1.5 * spread + 1e-3 * scale
    if verbose:
        print(f"smoke ok: max|hip-oracle_bf16|={err:.3e} (oracle bf16-vs-fp32 spread {spread:.3e}, scale {scale:.3e})")
    return err
