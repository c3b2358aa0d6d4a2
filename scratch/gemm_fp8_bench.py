import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops
for (M,N,K) in [(6400,12288,4096),(6400,22016,4096),(6400,4096,11008),(23200,6144,4096),(23200,28672,4096),(23200,4096,14336),(256608,3456,1152),(256608,4352,1152),(256608,1152,4352),(67848,3840,1280),(8192,8192,8192)]:
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    aq,asc=ops.quantize_fp8(a); wq,wsc=ops.quantize_fp8(w)
    def tm(fn,n=6):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)/n*1e-3
    tb=tm(lambda: ops.linear(a,w)); tf=tm(lambda: ops.linear_fp8(aq,asc,wq,wsc)); tq=tm(lambda: ops.quantize_fp8(a))
    print(f"{M:7d} {N:6d} {K:6d} bf16 {2*M*N*K/tb/1e12:7.1f} TF ({tb*1e6:8.1f} us) | fp8 {2*M*N*K/tf/1e12:7.1f} TF ({tf*1e6:8.1f} us) | quantise A {tq*1e6:7.1f} us | speedup incl. quantise {tb/(tf+tq):.2f}x", flush=True)
