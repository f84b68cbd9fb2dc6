"""GPU probe: how much of the GEMM K-loop time is operand-miss latency?  Same launch geometry with the weight
operand (ldw = 0: every W row aliases row 0 -> always cache-hot), the token operand (ldx = 0), or both made hot."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import graph_time
DEV = "cuda:0"; dtype = torch.bfloat16
for (name, M, N, K, cfg) in [("geglu", 2048, 10240, 1280, (128, 128, 1)), ("to_out", 2048, 1280, 1280, (64, 64, 1)),
                             ("ff.out", 2048, 1280, 5120, (64, 64, 1)), ("qk", 2048, 2560, 1280, (128, 128, 1)),
                             ("to_out", 2048, 1280, 1280, (6064, 160, 1)), ("to_out", 2048, 1280, 1280, (5064, 64, 1)),
                             ("ff.out", 2048, 1280, 5120, (4128, 64, 1)), ("ff.out", 2048, 1280, 5120, (6064, 160, 1)),
                             ("geglu", 2048, 10240, 1280, (9128, 320, 1)), ("geglu", 2048, 10240, 1280, (9256, 320, 1))]:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    r = {}
    for lab, kw in (("normal", {}), ("W hot", dict(ldw=0)), ("X hot", dict(ldx=0)), ("both hot", dict(ldw=0, ldx=0))):
        r[lab] = graph_time(lambda c: c.gemm(x, w, out=out, cfg=cfg, **kw), dtype, n=20, reps=3) * 1e3
    print(f"{name:7s} {M}x{N}x{K} {cfg}: " + "  ".join(f"{k} {v:6.1f} us" for k, v in r.items()), flush=True)
