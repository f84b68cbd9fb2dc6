import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import timeit
DEV="cuda:0"; dtype=torch.bfloat16
ctx=Ctx(DEV,dtype)
for (M,N,K) in [(2048,10240,1280),(8192,5120,2560),(2048,1280,5120)]:
    x=torch.randn(M,K,device=DEV).to(dtype); w=(torch.randn(N,K,device=DEV)*K**-0.5).to(dtype); out=torch.empty(M,N,device=DEV,dtype=dtype)
    for cfg in [(128,128,1),(64,64,1),(128,64,1)]:
        line=f"M={M} N={N} K={K} cfg={cfg}:"
        for name,fl in [("full",0),("noload",1<<8),("noread",1<<9),("nomfma",1<<10),("noload+noread",3<<8),("noread+nomfma",3<<9),("noload+nomfma",5<<8),("none",7<<8)]:
            ms=timeit(lambda: ctx.gemm(x,w,cfg=cfg,out=out,flags=fl))
            line+=f"  {name}={ms*1e3:.1f}us"
        print(line,flush=True)
