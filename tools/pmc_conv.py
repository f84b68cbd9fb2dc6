"""conv1 @ 128^2 (2 x 128 x 128 x 320 -> 320, the largest-M conv of the forward), tile variants, 4 launches each, for a
rocprofv3 --pmc FETCH_SIZE pass: fabric fetch per launch with and without the LDS halo.  python tools/pmc_conv.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
B, H, W, Cin, N = 2, 128, 128, 320, 320
x = torch.randn(B, H, W, Cin, device=DEV).to(dtype)
w = (torch.randn(N, 9 * Cin, device=DEV) * (9 * Cin) ** -0.5).to(dtype)
out = torch.empty(B, H, W, N, device=DEV, dtype=dtype)
for cfg in [(64, 128, 1), (256, 256, 1), (6128, 320, 1), (7128, 320, 1)]:
    for _ in range(4):
        ctx.conv3x3(x, w, cfg=cfg, out=out)
torch.cuda.synchronize()
