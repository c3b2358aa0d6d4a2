import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
M,N,K,which=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3]),int(sys.argv[4])
a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=(torch.randn(N,K,device='cuda')*0.02).to(torch.bfloat16)
_lib.lib().licv_gemm_select(which)
for _ in range(3): o=ops.linear(a,w)
torch.cuda.synchronize()
