"""GPU probe: tile-count quantisation of the 128x128 GEMM at M = 2048, K = 1280 (the FF GEGLU shape is N = 10240 =
1280 tiles = 5 per CU): time vs N.  If time per tile is flat across N there is nothing to win from tile shapes
that make the count a multiple of 256/512."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import graph_time
DEV = "cuda:0"; dtype = torch.bfloat16
M, K = 2048, 1280
x = torch.randn(M, K, device=DEV).to(dtype)
for flags, name in ((0, "plain"), (L.GF_GEGLU, "geglu")):
    for N in (4096, 6144, 8192, 9216, 10240, 12288, 16384):
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
        b = torch.randn(N, device=DEV).to(dtype)
        out = torch.empty(M, N // 2 if flags else N, device=DEV, dtype=dtype)
        t = graph_time(lambda c: c.gemm(x, w, bias=b, out=out, cfg=(128, 128, 1), flags=flags), dtype, n=20, reps=3)
        tiles = (M // 128) * (N // 128)
        print(f"{name} N={N:6d} tiles={tiles:5d} ({tiles/256:.2f}/CU) {t*1e3:7.1f} us  {t*1e6/tiles*256:6.2f} us per tile-per-CU  {2.0*M*N*K/t/1e9:6.0f} TF/s", flush=True)

print("--- fixed cost per round: 2048 x 8192 (4 tiles per CU = 2 rounds of 2) vs K, and vs tile shape at K = 64")
N = 8192
for K2 in (64, 128, 320, 640, 1280, 2560, 5120):
    x2 = torch.randn(M, K2, device=DEV).to(dtype); w2 = (torch.randn(N, K2, device=DEV) * K2 ** -0.5).to(dtype)
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    t = graph_time(lambda c: c.gemm(x2, w2, out=out, cfg=(128, 128, 1)), dtype, n=20, reps=3)
    print(f"K={K2:5d} {t*1e3:7.1f} us  ({K2//64} iterations)", flush=True)
x2 = torch.randn(M, 64, device=DEV).to(dtype)
for N2, cfg in ((8192, (64, 64, 1)), (1280, (64, 64, 1)), (1280, (128, 128, 1)), (2560, (128, 128, 1))):
    w2 = torch.randn(N2, 64, device=DEV).to(dtype); out = torch.empty(M, N2, device=DEV, dtype=dtype)
    t = graph_time(lambda c: c.gemm(x2, w2, out=out, cfg=cfg), dtype, n=20, reps=3)
    print(f"K=64 N={N2} cfg={cfg}: {t*1e3:7.1f} us ({(M//cfg[0])*(N2//cfg[1])} tiles)", flush=True)
t = graph_time(lambda c: c.ew(L.EW_STEP_SET, torch.zeros(1, dtype=torch.int32, device=DEV), i=(0, 0, 0, 0, 0, 0)), dtype, n=20, reps=3)
print(f"empty-ish kernel (step++): {t*1e3:.2f} us per graph node")
