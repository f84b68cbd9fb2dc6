"""GEGLU-shape GEMM (2048 x 10240 x 1280), a few tile variants, 4 launches each, for rocprofv3 --pmc passes
(FETCH_SIZE; TCC_HIT_sum TCC_MISS_sum): python tools/pmc_geglu.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
for (M, N, K) in [(2048, 10240, 1280), (2048, 1280, 1280)]:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    for cfg in [(128, 128, 1), (64, 64, 1), (5258, 320, 1), (9256, 320, 1), (8256, 256, 1)]:
        for _ in range(4):
            ctx.gemm(x, w, cfg=cfg, out=out)
    torch.cuda.synchronize()
