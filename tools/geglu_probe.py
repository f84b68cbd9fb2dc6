import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import graph_time
DEV="cuda:0"; dtype=torch.bfloat16
M,N,K=2048,10240,1280
x=torch.randn(M,K,device=DEV).to(dtype); w=(torch.randn(N,K,device=DEV)*K**-0.5).to(dtype); b=torch.randn(N,device=DEV).to(dtype)
res=torch.randn(M,N//2,device=DEV).to(dtype)
out=torch.empty(M,N,device=DEV,dtype=dtype); outh=torch.empty(M,N//2,device=DEV,dtype=dtype)
for cfg in [(128,128,1),(256,128,1),(128,64,1)]:
    t0=graph_time(lambda c: c.gemm(x,w,cfg=cfg,out=out),dtype)
    t1=graph_time(lambda c: c.gemm(x,w,bias=b,cfg=cfg,out=out),dtype)
    t2=graph_time(lambda c: c.gemm(x,w,bias=b,flags=L.GF_GEGLU,cfg=cfg,out=outh),dtype)
    t3=graph_time(lambda c: c.gemm(x,w,bias=b,flags=L.GF_ACT_GELU,cfg=cfg,out=out),dtype)
    print(f"cfg={cfg}: plain {t0*1e3:.1f}us  +bias {t1*1e3:.1f}us  +bias+GEGLU {t2*1e3:.1f}us  +bias+GELU(full N) {t3*1e3:.1f}us",flush=True)
# ff.out with residual
M,N,K=2048,1280,5120
x=torch.randn(M,K,device=DEV).to(dtype); w=(torch.randn(N,K,device=DEV)*K**-0.5).to(dtype); b=torch.randn(N,device=DEV).to(dtype); r=torch.randn(M,N,device=DEV).to(dtype); out=torch.empty(M,N,device=DEV,dtype=dtype)
for cfg in [(64,64,1),(3128,128,1)]:
    t0=graph_time(lambda c: c.gemm(x,w,cfg=cfg,out=out),dtype); t1=graph_time(lambda c: c.gemm(x,w,bias=b,residual=r,cfg=cfg,out=out),dtype)
    print(f"ff.out cfg={cfg}: plain {t0*1e3:.1f}us  +bias+residual {t1*1e3:.1f}us",flush=True)
