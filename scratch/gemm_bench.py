import sys, torch, time
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
shapes = [(6400,12288,4096),(6400,4096,4096),(6400,22016,4096),(6400,4096,11008),(67848,3840,1280),(67848,1280,1280),(67848,5120,1280),(67848,1280,5120),(6400,32002,4096),(16896,8192,1280),(4096,4096,4096),(8192,8192,8192)]
if len(sys.argv)>1: shapes=[tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
lib=_lib.lib()
for (M,N,K) in shapes:
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    res={}
    outs={}
    for which in (6,8,18,6,8,18):
        lib.licv_gemm_select(8 if which==18 else which); lib.licv_gemm_stagger(0 if which==18 else 1)
        for _ in range(2): o=ops.linear(a,w)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        n=10
        e0.record()
        for _ in range(n): o=ops.linear(a,w)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/n*1e-3
        res.setdefault(which,[]).append(2*M*N*K/t/1e12)
        outs[which]=o
    same=torch.equal(outs[6],outs[8])
    ref=(a[:64].float()@w.float().t())
    err=float((outs[6][:64].float()-ref).abs().max()/ref.abs().max())
    print(f"{M:6d} {N:6d} {K:6d}  pingpong {max(res[6]):7.1f} TF  persist+stagger {max(res[8]):7.1f}  persist-nostagger {max(res[18]):7.1f} TF  same={same} relerr={err:.2e}", flush=True)
lib.licv_gemm_select(0)
