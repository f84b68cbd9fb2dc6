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
