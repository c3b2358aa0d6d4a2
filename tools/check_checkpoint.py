#!/usr/bin/env python3
"""Real-checkpoint parity run (SURVEY.md §7 hard part 1): wherever a downloaded checkpoint directory exists
(ref:utils.py:42,70 — ``MODEL_CPK_DIR/<model_name>``, from ``huggingface-cli download HuggingFaceM4/idefics-9b`` or
``…/idefics2-8b-base``, ref:README.md:33-40), load it into the native engine through the drop-in interface, run a hooked forward
and a hooked 3-beam generate on a synthetic image+text batch, and compare with the CPU oracle on the SAME weights
(bf16, full depth; add --fp32 for the fp32 oracle as well, which needs 4 bytes/parameter of host RAM).

    python tools/check_checkpoint.py /path/to/idefics-9b [--shots 1] [--batch 1] [--fp32]

Exit code 0 = within the bar of tests/test_fullwidth_gpu.py (engine no less accurate than the oracle's bf16 path).  No
checkpoint exists in the build container or on the GPU box (no network), so this script is exercised there only on the tiny
save_pretrained directory of tests/test_dropin_gpu.py.
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--shots", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    from transformers import AutoConfig
    from licv.synthetic import synth_icv, synth_vqa_batch, synth_vqa_batch_idefics2
    from lmm_icl_interface import Idefics2Interface, IdeficsInterface
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    cfg = AutoConfig.from_pretrained(args.path)
    is2 = cfg.model_type == "idefics2"
    t0 = time.perf_counter()
    iface = (Idefics2Interface if is2 else IdeficsInterface)(args.path, "bf16", args.device)
    arch = iface.arch
    sd, _, _, _ = iface._load_checkpoint(Path(args.path), None, None, type(arch))
    print(f"loaded {args.path} ({cfg.model_type}, {arch.num_layers} layers, hidden {arch.hidden_size}) in {time.perf_counter() - t0:.1f}s")
    n_img = args.shots + 1
    if is2:
        from oracle import idefics2_ref as R
        fmt = "model.model.text_model.layers.<LAYER_NUM>.mlp"
        batch = synth_vqa_batch_idefics2(arch, args.batch, n_img * (arch.r_latents + 2) + 24, n_img, 378, 504, seed=426, dtype=torch.float32)
    else:
        from oracle import idefics_ref as R
        fmt = "model.model.layers.<LAYER_NUM>"
        batch = synth_vqa_batch(arch, args.batch, 24 * n_img + 8, n_img, seed=426, dtype=torch.float32, image_token_id=getattr(iface, "image_token_id", None))
    icv, alpha = synth_icv(arch.num_layers, arch.hidden_size, seed=426, alpha=0.1)
    icv_eff = alpha.unsqueeze(-1) * icv
    layers = list(range(arch.num_layers))
    w = LearnableICVInterventionLMM(iface, True, -1, fmt, arch.num_layers)
    dev_batch = {k: v.to(args.device) for k, v in batch.items()}
    with torch.no_grad():
        got = w(icv=icv_eff.to(args.device), **dev_batch)["logits"].float().cpu()
        ids = w.generate(icv=icv_eff.to(args.device), **dev_batch, max_new_tokens=5, num_beams=3, length_penalty=0.0).cpu()
    gold = {}
    for name, dt in (("bf16", torch.bfloat16),) + ((("f32", torch.float32),) if args.fp32 else ()):
        s = {k: v.to(dt) for k, v in sd.items()}
        kw = dict(batch)
        kw["pixel_values"] = batch["pixel_values"].to(dt)
        t1 = time.perf_counter()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=(is2 and dt == torch.bfloat16)):
            gold[name] = R.forward(s, arch, **kw, icv=icv_eff, hook_layers=layers).float()
        print(f"CPU oracle {name}: {time.perf_counter() - t1:.1f}s on {torch.get_num_threads()} threads")
    scale = float(gold["bf16"].abs().max())
    e16 = float((got - gold["bf16"]).abs().max())
    ok = e16 <= 1.5e-2 * scale
    print(f"max|native - oracle bf16| = {e16 / scale:.2e} of the logit scale {scale:.3g}")
    if args.fp32:
        spread = float((gold["bf16"] - gold["f32"]).abs().max())
        e32 = float((got - gold["f32"]).abs().max())
        print(f"max|native - oracle fp32| = {e32 / scale:.2e}; oracle bf16-vs-fp32 spread {spread / scale:.2e}")
        ok = e32 <= 1.5 * spread + 1e-3 * scale and e16 <= max(1.5e-2 * scale, 1.5 * spread)
    same = float((got.argmax(-1) == gold["bf16"].argmax(-1))[batch["attention_mask"].bool()].float().mean())
    print(f"next-token argmax agreement with the bf16 oracle on real tokens: {100 * same:.1f} %; generated ids {ids[:, -5:].tolist()}")
    print("PARITY OK" if ok else "PARITY FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
