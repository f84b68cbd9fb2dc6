"""How much do cold (HBM-resident) weights cost a GEMM vs warm (L2/MALL) ones?  Each shape is timed inside a
hipGraph cycling through R different weight buffers (R*|W| >> 256 MB MALL) vs re-using one buffer."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
DEV="cuda:0"; dtype=torch.bfloat16
def run(M,N,K,cfg,R):
    x=torch.randn(M,K,device=DEV).to(dtype); out=torch.empty(M,N,device=DEV,dtype=dtype)
    ws=[(torch.randn(N,K,device=DEV)*K**-0.5).to(dtype) for _ in range(R)]
    res=[]
    for mode in ("warm","cold"):
        rec=Ctx(DEV,dtype,record=True)
        n=max(R,24)
        for i in range(n): rec.gemm(x, ws[i%R] if mode=="cold" else ws[0], cfg=cfg, out=out)
        rec.capture(); rec.replay(); torch.cuda.synchronize()
        best=1e9
        for _ in range(3):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(); rec.replay(); e1.record(); torch.cuda.synchronize()
            best=min(best,e0.elapsed_time(e1)/n)
        res.append(best*1e3)
    return res
for (name,M,N,K,cfg) in [("to_q",2048,1280,1280,(64,64,1)),("to_qk",2048,2560,1280,(128,64,1)),("ff.geglu",2048,10240,1280,(128,128,1)),("ff.out",2048,1280,5120,(64,64,1))]:
    wbytes=N*K*2
    R=max(2,int(600e6//wbytes))
    R=min(R,96)
    w,c=run(M,N,K,cfg,R)
    print(f"{name:9s} W={wbytes/1e6:5.1f} MB x{R:3d}: warm {w:6.1f} us  cold {c:6.1f} us  (+{c-w:.1f})",flush=True)
