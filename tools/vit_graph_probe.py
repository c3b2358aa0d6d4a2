#!/usr/bin/env python3
"""How much of the vision tower's time at 8 images (the student pass, generate's prefill: M = 2056) is launch spacing and kernel ramp /
tail that a HIP graph or a second stream could take back?  Times IdeficsEngine.encode_images on 8 images: (a) eager (Python loop, ctypes
launches), (b) the same launches captured once in a HIP graph and replayed, (c) the two halves of the image batch captured on two streams
(fork / join inside the capture) and replayed.  Results of (b) and (c) are compared with (a) bit for bit."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv.config import IDEFICS_9B
from licv.idefics_engine import IdeficsEngine, IdeficsWeights
from licv.synthetic import synth_idefics_weights

dev = "cuda"
arch = IDEFICS_9B.with_(num_layers=4)                   # the full vision tower + perceiver; the text stack is not used here
sd = synth_idefics_weights(arch, seed=5, dtype=torch.bfloat16, device=dev)
eng = IdeficsEngine(IdeficsWeights(sd, arch, dev))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pix = torch.randn(n, 1, 3, arch.v_image, arch.v_image, device=dev, dtype=torch.bfloat16)

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

ref = eng.encode_images(pix).clone()
t_eager = timed(lambda: eng.encode_images(pix))
print(f"{n} images: eager {t_eager:.3f} ms", flush=True)

# (b) one graph
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): eng.encode_images(pix)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out_b = eng.encode_images(pix)
g.replay(); torch.cuda.synchronize()
print(f"  one graph: {timed(g.replay):.3f} ms, bit-identical to eager: {bool(torch.equal(out_b, ref))}", flush=True)

# (c) two halves on two streams inside one graph
h = n // 2
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def halves():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    a = eng.encode_images(pix[:h])
    with torch.cuda.stream(s2):
        b = eng.encode_images(pix[h:])
    cur.wait_stream(s2)
    return a, b
with torch.cuda.stream(s1):
    for _ in range(2): halves()
torch.cuda.synchronize()
t_halves_eager = timed(lambda: halves())
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    a, b = halves()
g2.replay(); torch.cuda.synchronize()
same = bool(torch.equal(torch.cat([a, b]), ref))
print(f"  two halves, eager on two streams: {t_halves_eager:.3f} ms; as one graph: {timed(g2.replay):.3f} ms, bit-identical to eager: {same}", flush=True)
