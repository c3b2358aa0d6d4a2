import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops
for (M,N,K) in [(256,12288,4096),(256,4096,4096),(256,22016,4096),(256,4096,11008),(256,32008,4096),(24,12288,4096),(24,22016,4096),(24,4096,11008),(848,6144,4096)]:
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    def tm(n=10):
        for _ in range(3): ops.linear(a,w)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): ops.linear(a,w)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)/n*1e3
    ops.SPLITK=False; t0=tm(); ops.SPLITK=True; t1=tm()
    wb=N*K*2/1e6
    print(f"{M:4d} {N:6d} {K:6d} plain {t0:7.1f} us  splitk {t1:7.1f} us (splits {ops._splitk_plan(M,N,K)[0]})  weights {wb:6.1f} MB -> {wb/t1*1e-3*1e3:6.2f} TB/s", flush=True)
