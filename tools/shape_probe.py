"""GPU: time the main SDXL GEMM shapes with whatever library IMH_LIB_PATH points to (A/B of kernel builds)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import graph_time
DEV = "cuda:0"; dtype = torch.bfloat16
c0 = Ctx(DEV, dtype)
out_line = []
for (name, M, N, K, conv) in [("geglu", 2048, 10240, 1280, None), ("to_out", 2048, 1280, 1280, None), ("ff.out", 2048, 1280, 5120, None),
                              ("qk", 2048, 2560, 1280, None), ("geglu64", 8192, 5120, 640, None), ("big", 8192, 5120, 2560, None),
                              ("conv64", 8192, 640, 5760, (2, 64, 64, 640)), ("conv32", 2048, 1280, 11520, (2, 32, 32, 1280))]:
    if conv:
        B, H, W, Cin = conv
        x = torch.randn(B, H, W, Cin, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
        out = torch.empty(B, H, W, N, device=DEV, dtype=dtype)
        f = lambda c: c.conv3x3(x, w, out=out)
    else:
        x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
        out = torch.empty(M, N, device=DEV, dtype=dtype)
        f = lambda c: c.gemm(x, w, out=out)
    t = min(graph_time(f, dtype, n=20, reps=3) for _ in range(2)) * 1e3
    out_line.append(f"{name} {t:.1f}")
print(os.path.basename(os.environ.get("IMH_LIB_PATH", "default")), " | ".join(out_line), flush=True)
