#!/usr/bin/env python3
"""Where do two runs of the Idefics2 fp8 (text + SigLIP) forward at B = 8 differ?  (tests/test_fullsize_gpu.py F1 at the bench's
own configuration.)  Prints, per stage, whether repeated / sliced runs are bit-identical."""
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
vl = int(sys.argv[2]) if len(sys.argv) > 2 else 27
arch = IDEFICS2_8B.with_(num_layers=nl, v_layers=vl)
sd = trained_like_(synth_idefics2_weights(arch, seed=426, dtype=torch.bfloat16, device=DEV), arch.num_layers)
e8 = Idefics2Engine(Idefics2Weights(sd, arch, DEV, fp8_text=True, fp8_vision=True))
del sd
B = 8
batch = synth_vqa_batch_idefics2(arch, B, 2900, 33, 378, 504, seed=426, min_len=2800, dtype=torch.bfloat16, device=DEV, ragged=False)
icv, alpha = synth_icv(arch.num_layers, arch.hidden_size, seed=426, alpha=0.1, device=DEV)
layers = list(range(arch.num_layers))
scaled = alpha.unsqueeze(-1) * icv
pv, pam = batch["pixel_values"], batch["pixel_attention_mask"]


def eq(a, b):
    if torch.equal(a, b):
        return "identical"
    d = (a.float() - b.float()).abs()
    return f"DIFFER: {int((d > 0).sum())} of {d.numel()} elements, max {float(d.max()):.3e}, rows touched {int((d.reshape(d.shape[0], -1).amax(1) > 0).sum())}"


img1 = e8.encode_images(pv, pam).clone()
img2 = e8.encode_images(pv, pam).clone()
print("vision tower + connector, all 264 images, run 1 vs run 2:", eq(img1, img2))
h1 = torch.cat([e8.encode_images(pv[:4], pam[:4]), e8.encode_images(pv[4:], pam[4:])]).clone()
print("vision tower, 2 x 132 images (sequential) vs 264 at once:", eq(img1, h1))
ins = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], image_hidden_states=img1)
t1 = e8.forward(**ins, icv=scaled, hook_layers=layers).clone()
t2 = e8.forward(**ins, icv=scaled, hook_layers=layers).clone()
print("text stack on fixed image states, whole batch, run 1 vs run 2:", eq(t1, t2))
cap = {}
t3 = e8.forward(**ins, icv=scaled, hook_layers=layers, capture=cap).clone()
del cap
print("text stack, capture path vs plain path:", eq(t1, t3))
ins4a = dict(input_ids=batch["input_ids"][:4], attention_mask=batch["attention_mask"][:4], image_hidden_states=img1[: img1.shape[0] // 2])
t4 = e8.forward(**ins4a, icv=scaled, hook_layers=layers).clone()
print("text stack, first 4 questions alone vs inside the batch of 8:", eq(t1[:4], t4))
s1 = e8.forward(**batch, icv=scaled, hook_layers=layers).clone()
s2 = e8.forward(**batch, icv=scaled, hook_layers=layers).clone()
print("full forward from pixels (2 batch streams), run 1 vs run 2:", eq(s1, s2))
print("full forward from pixels (2 streams) vs text stack on the 264-image states:", eq(s1, t1))
e8.batch_streams = 0
u1 = e8.forward(**batch, icv=scaled, hook_layers=layers).clone()
print("full forward, streams off, vs text stack on the 264-image states:", eq(u1, t1))
print("full forward, streams off vs 2 streams:", eq(u1, s1))
