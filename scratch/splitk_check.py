import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops
torch.manual_seed(0)
for (M,N,K) in [(172,4096,14336),(256,1024,11008),(24,4096,8192)]:
    a=torch.randn(M,K).bfloat16(); w=(torch.randn(N,K)*0.03).bfloat16()
    ref=(a.double()@w.double().T)
    ad,wd=a.cuda(),w.cuda()
    ops.SPLITK=True; y1=ops.linear(ad,wd).float().cpu().double()
    ops.SPLITK=False; y0=ops.linear(ad,wd).float().cpu().double()
    ops.SPLITK=True
    rel=lambda x,y: float((x-y).norm()/y.norm())
    print(M,N,K,"split-vs-plain",rel(y1,y0),"split-vs-ref",rel(y1,ref),"plain-vs-ref",rel(y0,ref), "bf16 rounding of ref", rel(ref.float().bfloat16().double(),ref))
