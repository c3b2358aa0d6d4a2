import sys, time; sys.path[:0]=['/root/repo','/root/repo/licv-vqa_amd']
import torch
from licv import ops
from licv.config import idefics_arch
from licv.idefics_engine import IdeficsEngine, IdeficsWeights
from licv.synthetic import synth_icv, synth_idefics_weights, synth_vqa_batch
dev=torch.device("cuda",0)
arch=idefics_arch("idefics-9b")
sd=synth_idefics_weights(arch, seed=0, device=dev, dtype=torch.bfloat16)
eng=IdeficsEngine(IdeficsWeights(sd, arch, dev)); del sd
batch=synth_vqa_batch(arch, 8, 800, 33, seed=426, min_len=700, dtype=torch.bfloat16, device=dev)
icv,alpha=synth_icv(arch.num_layers, arch.hidden_size, seed=426, alpha=0.1, device=dev)
hooks=dict(icv=icv, alpha=alpha, hook_layers=list(range(arch.num_layers)))
KEYS=("input_ids","attention_mask","pixel_values","image_attention_mask")
streams=[torch.cuda.Stream(device=dev) for _ in range(4)]
def split(parts):
    cur=torch.cuda.current_stream(dev); B=8; outs=[]
    bounds = parts if isinstance(parts, list) else [B*i//parts for i in range(parts+1)]
    for i in range(len(bounds)-1):
        st=streams[i]; st.wait_stream(cur)
        with torch.cuda.stream(st):
            sl=slice(bounds[i], bounds[i+1])
            outs.append(eng.forward(**{k:batch[k][sl] for k in KEYS}, **hooks))
    for i in range(len(bounds)-1): cur.wait_stream(streams[i])
    for o in outs: o.record_stream(cur)
    return torch.cat(outs,0)
def bench(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): out=fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3, out
t1,o1=bench(lambda: eng.forward(**batch, **hooks))
print("single stream", round(t1,1),"ms",flush=True)
for p in (2,[0,5,8],[0,3,6,8],[0,4,8]):
    tp,op=bench(lambda: split(p))
    print(p,"streams", round(tp,1),"ms equal", bool(torch.equal(o1,op)),flush=True)
t1,o1=bench(lambda: eng.forward(**batch, **hooks))
print("single stream", round(t1,1),"ms",flush=True)
