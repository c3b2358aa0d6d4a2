"""Per-shape GEMM time inside the headline forward (event pairs around each launch)."""
import sys, collections, torch
sys.path.insert(0, 'licv-vqa_amd'); sys.path.insert(0, '.')
from licv import ops
from licv.config import idefics_arch
from licv.idefics_engine import IdeficsEngine, IdeficsWeights
from licv.synthetic import synth_icv, synth_idefics_weights, synth_vqa_batch
dev = torch.device('cuda')
arch = idefics_arch('idefics-9b')
sd = synth_idefics_weights(arch, seed=426, dtype=torch.bfloat16, device=dev)
eng = IdeficsEngine(IdeficsWeights(sd, arch, dev)); del sd
batch = synth_vqa_batch(arch, 8, 800, 33, seed=426, min_len=720, dtype=torch.bfloat16, device=dev)
icv, alpha = synth_icv(arch.num_layers, arch.hidden_size, seed=426, alpha=0.1, device=dev)
hooks = dict(icv=icv, alpha=alpha, hook_layers=list(range(arch.num_layers)))
for _ in range(2): eng.forward(**batch, **hooks)
prof = []; ops.set_profiler(prof)
torch.cuda.synchronize()
import time; t0 = time.perf_counter()
for _ in range(3): eng.forward(**batch, **hooks)
torch.cuda.synchronize(); el = time.perf_counter() - t0
ops.set_profiler(None)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for rec in prof:
    if rec[0] != 'gemm': continue
    a = agg[rec[4]]; a[0] += 1; a[1] += rec[1].elapsed_time(rec[2]) * 1e-3; a[2] += rec[3]
tot = sum(a[1] for a in agg.values())
print(f"step {el/3*1e3:.1f} ms, gemm {tot/3*1e3:.1f} ms/step")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"M={k[0]:6d} N={k[1]:6d} K={k[2]:6d} calls/step {a[0]//3:4d} avg {a[1]/a[0]*1e6:8.1f} us  {a[2]/a[1]/1e12:7.1f} TF  share {100*a[1]/tot:5.1f}%")
