import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
lib=_lib.lib()
sels=[int(x) for x in sys.argv[1].split(',')]
cases=[(67848,3840,1280),(67848,1280,5120),(6400,22016,4096),(6400,12288,4096),(6400,4096,11008),(8192,8192,8192)]
for (M,N,K) in cases:
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    line=f"{M:6d} {N:6d} {K:6d}"
    for sel in sels+sels:
        lib.licv_gemm_select(sel)
        for _ in range(2): o=ops.linear(a,w)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        n=8
        e0.record()
        for _ in range(n): o=ops.linear(a,w)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/n*1e-3
        line+=f" | sel{sel}: {2*M*N*K/t/1e12:7.1f} TF"
    print(line, flush=True)
lib.licv_gemm_select(0)
