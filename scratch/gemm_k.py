import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
lib=_lib.lib()
M,N=67848,3840
for K in (128,256,512,1280,2560):
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    for which,name in ((6,'pingpong'),(7,'pp-noepi'),(2,'tile256'),(1,'tile128')):
        lib.licv_gemm_select(which)
        for _ in range(2): o=ops.linear(a,w)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): o=ops.linear(a,w)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/10*1e-3
        print(f"K={K:5d} {name:9s} {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF", flush=True)
# pure device copy of the same output volume for reference
x=torch.empty(M,N,device='cuda',dtype=torch.bfloat16); y=torch.empty_like(x)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize()
print("copy 521MB:", e0.elapsed_time(e1)/10*1e3, "us")
e0.record()
for _ in range(10): y.zero_()
e1.record(); torch.cuda.synchronize()
print("memset 521MB:", e0.elapsed_time(e1)/10*1e3, "us")
lib.licv_gemm_select(0)
