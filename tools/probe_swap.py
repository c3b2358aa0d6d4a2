import sys; sys.path[:0]=['/root/repo','/root/repo/licv-vqa_amd']
import torch, ctypes as C
from licv import _lib
out=torch.zeros(128,dtype=torch.int32,device='cuda')
_lib.check(_lib.lib().licv_probe_permlane16_swap(C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
o=out.cpu().view(64,2)
for l in (0,1,15,16,17,31,32,33,47,48,63): print(l, o[l].tolist())
