import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import timeit
DEV="cuda:0"; dtype=torch.bfloat16
ctx=Ctx(DEV,dtype)
variants=[(128,128,1),(256,128,1),(256,256,1),(3128,128,1)]
for (M,N,K) in [(8192,5120,2560),(8192,8192,8192),(4096,4096,4096),(2048,10240,1280),(8192,5120,640)]:
    x=torch.randn(M,K,device=DEV).to(dtype); w=(torch.randn(N,K,device=DEV)*K**-0.5).to(dtype); out=torch.empty(M,N,device=DEV,dtype=dtype)
    line=f"M={M} N={N} K={K}:"
    for cfg in variants:
        ms=timeit(lambda: ctx.gemm(x,w,cfg=cfg,out=out),n=5)
        line+=f"  [{cfg[0]}x{cfg[1]} {ms*1e3:.1f}us {2.0*M*N*K/ms/1e9:.0f}TF]"
    print(line,flush=True)
