import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops
from licv.config import IDEFICS2_8B
from licv.idefics2_engine import Idefics2Engine, Idefics2Weights
from licv.synthetic import synth_idefics2_weights, synth_vqa_batch_idefics2, synth_icv
arch=IDEFICS2_8B
sd=synth_idefics2_weights(arch,seed=426,dtype=torch.bfloat16,device='cuda')
eng=Idefics2Engine(Idefics2Weights(sd,arch,'cuda')); del sd
batch=synth_vqa_batch_idefics2(arch,1,172,2,378,504,seed=426,min_len=160,dtype=torch.bfloat16,device='cuda',ragged=True)
icv,alpha=synth_icv(arch.num_layers,arch.hidden_size,seed=426,alpha=0.1,device='cuda')
layers=list(range(arch.num_layers)); scaled=alpha.unsqueeze(-1)*icv
caps=[]
for flag in (True,False,False):
    ops.SPLITK=flag; cap={}; lg=eng.forward(**batch,icv=scaled,hook_layers=layers,capture=cap); caps.append((cap,lg.float()))
ops.SPLITK=True
print("plain vs plain logits equal:", torch.equal(caps[1][1],caps[2][1]))
for l in (0,1,2,4,8,16,31):
    a=caps[0][0]["layer_out"][l].float(); b=caps[1][0]["layer_out"][l].float()
    m0=caps[0][0]["mlp_raw"][l].float(); m1=caps[1][0]["mlp_raw"][l].float()
    print(l, "layer_out rel", float((a-b).norm()/b.norm()), "mlp_raw rel", float((m0-m1).norm()/m1.norm()), "mlp max", float((m0-m1).abs().max()), float(m1.abs().max()))
print("logits rel", float((caps[0][1]-caps[1][1]).norm()/caps[1][1].norm()))
