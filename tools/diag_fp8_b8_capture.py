#!/usr/bin/env python3
"""The Idefics2 fp8 text stack's CAPTURE path at B = 8 (M = 23200 rows): which question / layer / tensor differs from the same question
run alone?  (tools/diag_fp8_b8.py found one question of eight differing between the capture path and the plain path.)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv.config import IDEFICS2_8B
from licv.idefics2_engine import Idefics2Engine, Idefics2Weights
from licv.synthetic import synth_icv, synth_idefics2_weights, synth_vqa_batch_idefics2, trained_like_

DEV = "cuda"
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
n_img_rows = img.shape[0] // B
capA = {}
lgA = e8.forward(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], image_hidden_states=img, icv=scaled, hook_layers=layers, capture=capA).clone()
plain = e8.forward(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], image_hidden_states=img, icv=scaled, hook_layers=layers).clone()
for q in range(B):
    capq = {}
    lgq = e8.forward(input_ids=batch["input_ids"][q:q + 1], attention_mask=batch["attention_mask"][q:q + 1],
                     image_hidden_states=img[q * n_img_rows:(q + 1) * n_img_rows], icv=scaled, hook_layers=layers, capture=capq)
    same = torch.equal(lgA[q], lgq[0])
    print(f"question {q}: capture(B=8)[q] vs capture(alone): {'identical' if same else 'DIFFER'}; plain(B=8)[q] vs capture(alone): "
          f"{'identical' if torch.equal(plain[q], lgq[0]) else 'DIFFER'}")
    if not same:
        for l in range(nl):
            for key in ("mlp_raw", "layer_out"):
                a, b = capA[key][l][q], capq[key][l][0]
                if not torch.equal(a, b):
                    d = (a.float() - b.float()).abs()
                    rows = (d.amax(-1) > 0).nonzero().flatten()
                    print(f"   layer {l} {key}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}, rows {rows[:6].tolist()}..{rows[-3:].tolist()} "
                          f"({rows.numel()} rows; flat row of the first = {q * 2900 + int(rows[0])})")
                    break
            else:
                continue
            break
