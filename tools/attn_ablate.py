import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import timeit
DEV="cuda:0"; dtype=torch.bfloat16
ctx=Ctx(DEV,dtype)
for (B,H,L) in [(8,10,4096),(2,20,1024)]:
    C_=H*64
    qk=torch.randn(B*L,2*C_,device=DEV).to(dtype); vt=torch.randn(C_,B*L,device=DEV).to(dtype); out=torch.empty(B*L,C_,device=DEV,dtype=dtype)
    line=f"self B={B} H={H} L={L}:"
    for name,abl in [("full",0),("noQK",1),("noSM",2),("noPV",4),("noload",8),("noQK+noPV",5),("noSM+noload",10),("onlySM",13),("onlyMFMA",10),("none",15)]:
        ctx.lib.imh_debug_set(1,abl)
        ms=timeit(lambda: ctx.attention(qk[:, :C_], qk[:, C_:], vt, out, B, H, L, L, L, 2*C_, 2*C_, B*L, C_, 0.125))
        line+=f"  {name}={ms*1e3:.0f}us"
    ctx.lib.imh_debug_set(1,0)
    print(line,flush=True)
