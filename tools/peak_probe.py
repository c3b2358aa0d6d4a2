"""Build-owned roofline probes: MFMA register loop (bf16 16x16x32) and HBM stream copy."""
import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import _lib
lib=_lib.lab()          # roofline probes live in liblicv_hip_lab.so
sink=torch.zeros(4,device='cuda')
st=torch.cuda.current_stream().cuda_stream
def tm(fn,n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
for waves_per_simd in (1,2,4):
    blocks=256*waves_per_simd; iters=20000
    t=tm(lambda: lib.licv_probe_mfma_loop(sink.data_ptr(),blocks,iters,st))
    fl=blocks*4*iters*8*16384
    print(f"MFMA loop, {waves_per_simd} wave(s)/SIMD: {fl/t/1e12:7.1f} TFLOP/s ({t*1e3:.1f} ms)")
for mb in (256, 2048, 8192):
    x=torch.empty(mb*1024*1024//4,device='cuda'); y=torch.empty_like(x)
    t=tm(lambda: y.copy_(x))
    print(f"stream copy {mb:5d} MiB: {2*x.numel()*4/t/1e9:7.0f} GB/s")
