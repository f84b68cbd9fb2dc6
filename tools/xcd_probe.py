import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import graph_time
DEV="cuda:0"; dtype=torch.bfloat16
ctx=Ctx(DEV,dtype)
for (name,M,N,K,cfg) in [("geglu",2048,10240,1280,(128,128,1)),("geglu640",8192,5120,640,(128,128,1)),("to_q",2048,1280,1280,(64,64,1)),("ff.out",2048,1280,5120,(64,64,1)),("qk",2048,2560,1280,(128,64,1))]:
    x=torch.randn(M,K,device=DEV).to(dtype); w=(torch.randn(N,K,device=DEV)*K**-0.5).to(dtype); out=torch.empty(M,N,device=DEV,dtype=dtype)
    line=f"{name:9s}"
    for mode,lab in [(5,"1x8"),(0,"auto"),(1,"legacy"),(5,"1x8"),(0,"auto"),(4,"2x4"),(0,"auto")]:
        ctx.lib.imh_debug_set(2,mode)
        line+=f"  {lab}={graph_time(lambda c: c.gemm(x,w,cfg=cfg,out=out),dtype)*1e3:.1f}us"
    ctx.lib.imh_debug_set(2,0)
    print(line,flush=True)
