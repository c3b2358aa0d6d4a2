import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
lib=_lib.lib()
# (M,N,K, epilogue)
cases=[(67848,5120,1280,'gelu'),(67848,1280,5120,'res16'),(67848,3840,1280,'bias'),(67848,1280,1280,'res16'),
       (6400,22016,4096,'swiglu'),(6400,4096,11008,'res32'),(6400,12288,4096,'none'),(6400,4096,4096,'res32')]
for (M,N,K,epi) in cases:
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    bias=(torch.randn(N,device='cuda')*0.1).to(torch.bfloat16)
    n_out = N//2 if epi=='swiglu' else N
    r16=torch.randn(M,n_out,device='cuda').to(torch.bfloat16); r32=torch.randn(M,n_out,device='cuda')
    kw={'gelu':dict(bias=bias,act='gelu'),'bias':dict(bias=bias),'res16':dict(bias=bias,residual=r16),'res32':dict(residual=r32),
        'swiglu':dict(swiglu=True),'none':{}}[epi]
    line=f"{M:6d} {N:6d} {K:6d} {epi:7s}"
    base=None
    for skew in (8,4,2,16,32,8):
        lib.licv_gemm_experiment(1, skew)   # knob 1 = patch group height (knob 0 = start stagger %)
        for _ in range(2): o=ops.linear(a,w,**kw)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        n=8
        e0.record()
        for _ in range(n): o=ops.linear(a,w,**kw)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/n*1e-3
        if base is None: base=o.float()
        err=float((o.float()-base).abs().max()/base.abs().max())
        line+=f" | skew{skew}: {2*M*N*K/t/1e12:7.1f} TF ({t*1e6:7.1f} us, d={err:.1e})"
    print(line, flush=True)
lib.licv_gemm_experiment(1, 0)
