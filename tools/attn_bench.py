import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import timeit
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV="cuda:0"; dtype=torch.bfloat16
ctx=Ctx(DEV,dtype)
def make_vt(v, n_pad):
    B, n, C_ = v.shape
    vt = v.permute(2, 0, 1).reshape(C_, B * n_pad // 16, 4, 4)
    return vt[:, :, [0, 2, 1, 3], :].reshape(C_, B * n_pad).contiguous()
for (B,H,L) in [(2,10,4096),(2,20,1024),(8,10,4096),(8,20,1024)]:
    C_=H*64
    qk=torch.randn(B*L,2*C_,device=DEV).to(dtype); v=torch.randn(B,L,C_,device=DEV).to(dtype); vt=make_vt(v,L); out=torch.empty(B*L,C_,device=DEV,dtype=dtype)
    line=f"self B={B} H={H} L={L}:"
    ref=None
    for nw in (4,):
        f=lambda: ctx.attention(qk[:, :C_], qk[:, C_:], vt, out, B, H, L, L, L, 2*C_, 2*C_, B*L, C_, 0.125)
        ms=timeit(f)
        if ref is None: ref=out.clone()
        ok=torch.equal(ref,out)
        line+=f"  [nw={nw} {ms*1e3:.1f}us {4.0*B*H*L*L*64/ms/1e9:.0f}TF same={ok}]"
    print(line,flush=True)
# cross attention 77 keys
for (B,H,L) in [(2,20,1024),(2,10,4096)]:
    C_=H*64
    q=torch.randn(B*L,C_,device=DEV).to(dtype); k=torch.zeros(B,128,C_,device=DEV,dtype=dtype); k[:,:77]=torch.randn(B,77,C_,device=DEV).to(dtype)
    v=torch.zeros(B,128,C_,device=DEV,dtype=dtype); v[:,:77]=torch.randn(B,77,C_,device=DEV).to(dtype); vt=make_vt(v,128); out=torch.empty(B*L,C_,device=DEV,dtype=dtype)
    line=f"cross B={B} H={H} L={L}:"
    for nw in (4,):
        ms=timeit(lambda: ctx.attention(q,k,vt,out,B,H,L,77,128,C_,C_,B*128,C_,0.125))
        line+=f"  [nw={nw} {ms*1e3:.1f}us {2*B*L*C_*2*2/ms/1e6:.0f}GB/s]"
    print(line,flush=True)
