import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
lib=_lib.lib()
for (M,N,K) in [(6400,12288,4096),(67848,3840,1280),(8192,8192,8192)]:
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    for which,name in ((6,'pingpong'),(7,'pp-noepi'),(6,'pingpong'),(7,'pp-noepi')):
        lib.licv_gemm_select(which)
        for _ in range(2): o=ops.linear(a,w)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): o=ops.linear(a,w)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/10*1e-3
        print(f"{M} {N} {K} {name:8s} {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF-equiv", flush=True)
lib.licv_gemm_select(0)
