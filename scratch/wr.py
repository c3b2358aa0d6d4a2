import sys, torch, ctypes as C
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import _lib
lib=_lib.lib()
lib.licv_dbg_tile_write.argtypes=[C.c_void_p,C.c_int64,C.c_int64,C.c_int64,C.c_int,C.c_int,C.c_void_p]
for (M,N) in ((67848,3840),(6400,12288)):
    x=torch.empty(M,N,device='cuda',dtype=torch.bfloat16)
    st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for mode,lds in ((0,0),(0,81920),(0,163840),(2,163840),(0,0)):
        for _ in range(2): assert lib.licv_dbg_tile_write(x.data_ptr(),N,M,N,mode,lds,st)==0
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): lib.licv_dbg_tile_write(x.data_ptr(),N,M,N,mode,lds,st)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/10*1e-3
        print(f"M={M} N={N} mode={mode} lds={lds}: {t*1e6:7.1f} us  {M*N*2/t/1e12:5.2f} TB/s", flush=True)
