"""Per-phase wall-clock breakdown of the GEMM tiles (100 MHz counter): select 6 = ping-pong kernel, 26 = lean kernel (which also
stamps the shader clock around its main loop: cycles per K stage and the clock the loop held).
usage: gemm_phases.py [--select 6|26] [M,N,K,epi ...]"""
import sys, ctypes, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
lib=_lib.lib()
_lib.lab()          # experiment kernels (csrc/lab/) register themselves with licv_gemm_select
cases=[(67848,5120,1280,'gelu'),(67848,1280,5120,'res16'),(67848,3840,1280,'bias'),(67848,1280,1280,'res16'),
       (6400,22016,4096,'swiglu'),(6400,4096,11008,'res32'),(6400,12288,4096,'none'),(6400,4096,4096,'res32')]
SEL=6
if len(sys.argv)>2 and sys.argv[1]=='--select': SEL=int(sys.argv[2]); del sys.argv[1:3]
if len(sys.argv)>1: cases=[tuple(int(x) if x.isdigit() else x for x in a.split(',')) for a in sys.argv[1:]]
for (M,N,K,epi) in cases:
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
    bias=(torch.randn(N,device='cuda')*0.1).to(torch.bfloat16)
    n_out=N//2 if epi=='swiglu' else N
    r16=torch.randn(M,n_out,device='cuda').to(torch.bfloat16); r32=torch.randn(M,n_out,device='cuda')
    kw={'gelu':dict(bias=bias,act='gelu'),'bias':dict(bias=bias),'res16':dict(bias=bias,residual=r16),'res32':dict(residual=r32),
        'swiglu':dict(swiglu=True),'none':{}}[epi]
    lib.licv_gemm_select(SEL)
    for _ in range(3): ops.linear(a,w,**kw)
    nt=((M+255)//256)*((N+255)//256)
    ts=torch.zeros(nt*8,dtype=torch.int64,device='cuda')
    assert lib.licv_gemm_debug_timestamps(ts.data_ptr())==0
    ops.linear(a,w,**kw); torch.cuda.synchronize()
    assert lib.licv_gemm_debug_timestamps(None)==0
    lib.licv_gemm_select(0)
    t=ts.view(nt,8).cpu().double()*0.01   # us
    t0=t[:,0].min()
    ph=[(t[:,i+1]-t[:,i]) for i in range(4)]
    tot=t[:,4]-t[:,0]
    print(f"{M:6d} {N:6d} {K:6d} {epi:7s} tiles {nt:5d} kernel span {float(t[:,4].max()-t0):8.1f} us | per tile: fill {float(ph[0].mean()):6.1f}  main {float(ph[1].mean()):6.1f}  phaseA {float(ph[2].mean()):5.1f}  phaseB {float(ph[3].mean()):6.1f}  total {float(tot.mean()):6.1f} (min {float(tot.min()):6.1f} max {float(tot.max()):6.1f}) | sum(tile)/256/span {float(tot.sum())/256/float(t[:,4].max()-t0):.2f}", flush=True)
    if SEL==26:
        raw=ts.view(nt,8).cpu().double()
        cyc=(raw[:,6]-raw[:,5]); wall=(raw[:,2]-raw[:,1])*10.0        # ns
        print(f"      main loop: {float(cyc.median())/(K//32):7.1f} shader cycles per 32-deep stage (median tile), clock {float((cyc/wall).median()):.2f} GHz")
    # start-time histogram: how many rounds
    st=((t[:,0]-t0)/float(tot.mean())).round()
    print("      starts per round:", [int((st==r).sum()) for r in range(int(st.max())+1)][:12])
