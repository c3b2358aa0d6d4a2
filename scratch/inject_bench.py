import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops
rows,H=6400,4096
h=torch.randn(rows,H,device='cuda'); v=torch.randn(H,device='cuda')*0.01; w=torch.ones(H,device='cuda',dtype=torch.bfloat16)
al=torch.tensor([0.1],device='cuda')
def tm(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
out=torch.empty_like(h)
t=tm(lambda: ops.inject_renorm(h,v,alpha=al,out=out,norm_weight=w,norm_eps=1e-6))
print(f"fused fp32->fp32+bf16: {t*1e6:.1f} us  {rows*H*10/t/1e9:.0f} GB/s")
t=tm(lambda: ops.inject_renorm(h,v,alpha=al,out=out))
print(f"plain fp32->fp32:      {t*1e6:.1f} us  {rows*H*8/t/1e9:.0f} GB/s")
hb=h.bfloat16()
t=tm(lambda: ops.inject_renorm(hb,v,alpha=al,norm_weight=w,norm_eps=1e-6))
print(f"fused bf16->fp32+bf16: {t*1e6:.1f} us  {rows*H*8/t/1e9:.0f} GB/s")
t=tm(lambda: ops.rmsnorm(h,w,1e-6))
print(f"rmsnorm fp32->bf16:    {t*1e6:.1f} us  {rows*H*6/t/1e9:.0f} GB/s")
x=torch.empty_like(h); 
t=tm(lambda: x.copy_(h))
print(f"torch copy fp32:       {t*1e6:.1f} us  {rows*H*8/t/1e9:.0f} GB/s")
