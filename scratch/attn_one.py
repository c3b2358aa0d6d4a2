import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
lib=_lib.lib()
name=sys.argv[1]; flags=int(sys.argv[2]) if len(sys.argv)>2 else 0
shapes={"vit":(264,257,257,16,16,80,0),"siglip":(16,972,972,16,16,72,0),"lm":(8,800,800,32,32,128,1)}
B,Sq,Sk,nh,nkv,hd,mode=shapes[name]
H=nh*hd; Hk=nkv*hd
q=torch.randn(B,Sq,H,device='cuda').bfloat16(); kv=torch.randn(B,Sk,2*Hk,device='cuda').bfloat16()
kw={}
if mode==1: kw['key_valid']=torch.ones(B,Sk,dtype=torch.int32,device='cuda')
lib.licv_attn_select(flags)
for _ in range(3): o=ops.attention(q,kv,kv.view(-1)[Hk:],B,Sq,Sk,nh,nkv,hd,Sq*H,H,Sk*2*Hk,2*Hk,hd**-0.5,mode,**kw)
torch.cuda.synchronize()
