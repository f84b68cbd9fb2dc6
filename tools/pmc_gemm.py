"""Run a few GEMM variants once each (for rocprofv3 --pmc passes): python tools/pmc_gemm.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
DEV="cuda:0"; dtype=torch.bfloat16
ctx=Ctx(DEV,dtype)
M,N,K=8192,5120,2560
x=torch.randn(M,K,device=DEV).to(dtype); w=(torch.randn(N,K,device=DEV)*K**-0.5).to(dtype); out=torch.empty(M,N,device=DEV,dtype=dtype)
from imagharmony_amd import lib as L
for cfg in [c for c in [(128,128,1),(256,128,1),(256,256,1),(64,64,1),(5258,320,1),(6128,320,1),(8256,256,1),(9128,320,1),(9256,320,1),
                        (2464,160,1),(24128,160,1),(23256,160,1)] if L.variant_built(c)]:      # (the experimental-only variants need IMH_LIB_PATH)
    for _ in range(3): ctx.gemm(x,w,cfg=cfg,out=out)
torch.cuda.synchronize()
# attention too
B,H,L=2,10,4096; C_=H*64
qk=torch.randn(B*L,2*C_,device=DEV).to(dtype); vt=torch.randn(C_,B*L,device=DEV).to(dtype); o=torch.empty(B*L,C_,device=DEV,dtype=dtype)
for _ in range(3): ctx.attention(qk[:, :C_], qk[:, C_:], vt, o, B, H, L, L, L, 2*C_, 2*C_, B*L, C_, 0.125)
torch.cuda.synchronize()
