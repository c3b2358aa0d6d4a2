import sys, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv import ops, _lib
lib=_lib.lib()
def run(name,B,Sq,Sk,nh,nkv,hd,mode,flags_list=(0,2,1,3)):
    H=nh*hd; Hk=nkv*hd
    q=torch.randn(B,Sq,H,device='cuda').bfloat16(); kv=torch.randn(B,Sk,2*Hk,device='cuda').bfloat16()
    kvld=torch.ones(B,Sk,dtype=torch.int32,device='cuda'); kvld[:, Sk-5:]=0
    kw={}
    if mode in (1,2): kw['key_valid']=kvld
    if mode==3:
        n_img=Sk//64; im=torch.zeros(B,Sq,n_img,dtype=torch.int32,device='cuda')
        idx=(torch.arange(Sq,device='cuda')*n_img//Sq).clamp(max=n_img-1)
        im[:,torch.arange(Sq),idx]=1
        kw.update(img_mask=im,img_len=64)
    fl=4.0*B*nh*Sq*Sk*hd*(0.5 if mode==1 else 1.0)
    line=f"{name:10s}"; ref=None
    for f in flags_list:
        lib.licv_attn_select(f)
        for _ in range(2): o=ops.attention(q,kv,kv.view(-1)[Hk:],B,Sq,Sk,nh,nkv,hd,Sq*H,H,Sk*2*Hk,2*Hk,hd**-0.5,mode,**kw)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): o=ops.attention(q,kv,kv.view(-1)[Hk:],B,Sq,Sk,nh,nkv,hd,Sq*H,H,Sk*2*Hk,2*Hk,hd**-0.5,mode,**kw)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/5*1e-3
        if ref is None: ref=o.float()
        d=float((o.float()-ref).abs().max())
        line+=f" | f{f}: {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF d={d:.0e}"
    print(line,flush=True)
    lib.licv_attn_select(0)
run("vit",264,257,257,16,16,80,0,(0,1))
run("lm",8,800,800,32,32,128,1,(0,))
run("siglip",16,972,972,16,16,72,0,(0,))
run("siglip-m",16,972,972,16,16,72,2,(0,))
run("perceiver",264,64,321,16,16,96,0,(0,1))
run("xattn",8,800,33*64,32,32,128,3,(0,))
run("mistral",8,172,172,32,8,128,1,(0,))
