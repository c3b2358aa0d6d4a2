#!/usr/bin/env python3
"""Idefics2 fp8 text stack at B = 8 (M = 23200): the product's fused kernel path against the unfused (capture) path, layer by layer.
tools/diag_fp8_b8*.py found ONE question of eight whose logits differ between the two paths at full depth although each path is
deterministic and batch-independent; this prints the first layer / rows / columns where the streams part."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv.config import IDEFICS2_8B
from licv.idefics2_engine import Idefics2Engine, Idefics2Weights
from licv.synthetic import synth_icv, synth_idefics2_weights, synth_vqa_batch_idefics2, trained_like_

DEV = "cuda"
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 32
arch = IDEFICS2_8B.with_(num_layers=nl, v_layers=2)
sd = trained_like_(synth_idefics2_weights(arch, seed=426, dtype=torch.bfloat16, device=DEV), 32)
e8 = Idefics2Engine(Idefics2Weights(sd, arch, DEV, fp8_text=True, fp8_vision=True))
del sd
B = 8
batch = synth_vqa_batch_idefics2(arch, B, 2900, 33, 378, 504, seed=426, min_len=2800, dtype=torch.bfloat16, device=DEV, ragged=False)
icv, alpha = synth_icv(arch.num_layers, arch.hidden_size, seed=426, alpha=0.1, device=DEV)
layers = list(range(arch.num_layers))
scaled = alpha.unsqueeze(-1) * icv
img = e8.encode_images(batch["pixel_values"], batch["pixel_attention_mask"])
ins = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], image_hidden_states=img, icv=scaled, hook_layers=layers)
capU = {}
e8.forward(**ins, capture=capU)
outs_u = [t.cpu() for t in capU["layer_out"]]
del capU
e8.capture_keeps_path = True
capF = {}
e8.forward(**ins, capture=capF)
for l in range(nl):
    a, b = outs_u[l], capF["layer_out"][l].cpu()
    if not torch.equal(a, b):
        d = (a.float() - b.float()).abs()
        rows = (d.amax(-1) > 0).nonzero()
        print(f"layer {l}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e} (scale {float(a.abs().max()):.3e}); "
              f"{rows.shape[0]} (question, position) rows: first {rows[:5].tolist()}")
        r0 = rows[0].tolist()
        cols = (d[r0[0], r0[1]] > 0).nonzero().flatten()
        print(f"   first row (q {r0[0]}, pos {r0[1]}, flat {r0[0] * 2900 + r0[1]}): {cols.numel()} columns differ, first {cols[:8].tolist()}; "
              f"values unfused {a[r0[0], r0[1], cols[:4]].tolist()} fused {b[r0[0], r0[1], cols[:4]].tolist()}")
        if rows.shape[0] > 64:
            break
else:
    print("no layer output differs")
