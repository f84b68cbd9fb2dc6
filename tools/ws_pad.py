"""Leading-dimension padding of both operands under the wave-specialised GEMM (is the LDS-DMA stream L2-channel bound?)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import graph_time
DEV = "cuda:0"; dtype = torch.bfloat16
for (M, N, K, cfg) in [(2048, 1280, 1280, (2464, 160, 1)), (2048, 1280, 5120, (2464, 160, 1)), (2048, 1280, 2560, (2464, 160, 1)),
                       (2048, 1280, 5120, (4128, 64, 1)), (2048, 10240, 1280, (9128, 320, 1)), (2048, 3840, 1280, (128, 128, 1))]:
    line = f"M={M} N={N} K={K} cfg={cfg}:"
    for pad in (0, 64, 128, 192, 320):
        xb = torch.randn(M, K + pad, device=DEV).to(dtype); wb = (torch.randn(N, K + pad, device=DEV) * K ** -0.5).to(dtype)
        x = xb[:, :K]; w = wb[:, :K]; out = torch.empty(M, N, device=DEV, dtype=dtype)
        us = graph_time(lambda c: c.gemm(x, w, cfg=cfg, out=out), dtype, n=20) * 1e3
        line += f"  pad{pad}={us:.1f}us"
    print(line, flush=True)
