import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
lib=_lib.lib()
for (M,N,K) in [(1376,6144,4096),(1376,4096,4096),(1376,28672,4096),(1376,4096,14336),(1376,32003,4096),(800,12288,4096),(800,4096,4096),(800,4096,11008),(15552,3456,1152),(15552,1152,4352),(15552,4352,1152),(15552,1152,1152)]:
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    line=f"{M:6d} {N:6d} {K:6d}"
    for sel in (0,1,2):
        lib.licv_gemm_select(0 if sel==2 else sel); ops.SPLITK = (sel==2)
        for _ in range(3): ops.linear(a,w)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.linear(a,w)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/10*1e-3
        line+=f" | {('pingpong','tile128 ','auto+splitk')[sel]}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:7.1f} TF"
    print(line,flush=True)
lib.licv_gemm_select(0); ops.SPLITK=True
