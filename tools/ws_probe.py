"""Wave-specialised 64 x 160 GEMM / conv (variant codes 1464 / 2464 / 24128) against the tuned variants on the M = 2048,
N = 1280 shapes: GPU-side (hipGraph) time per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import graph_time
DEV = "cuda:0"; dtype = torch.bfloat16
L.load()
GEMMS = [("to_out", 2048, 1280, 1280, [(64, 64, 1), (5064, 64, 1)]), ("ff.out", 2048, 1280, 5120, [(4128, 64, 1)]),
         ("shortcut", 2048, 1280, 2560, [(4128, 64, 1)]), ("shortcut", 2048, 1280, 640, [(64, 64, 1)]),
         ("qkv-ish", 2048, 3840, 1280, [(128, 128, 1), (128, 64, 1)]), ("to_out64", 8192, 640, 640, [(64, 64, 1), (128, 64, 1)]),
         ("ff.out64", 8192, 640, 2560, [(128, 64, 1), (64, 128, 1)]), ("geglu64", 8192, 5120, 640, [(128, 128, 1)]),
         ("proj64", 8192, 640, 1280, [(5064, 64, 1)]), ("geglu", 2048, 10240, 1280, [(9128, 320, 1)])]
WS = [(1464, 160, 1), (2464, 160, 1), (24128, 160, 1), (24128, 128, 1)]
for name, M, N, K, base in GEMMS:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
    b = torch.randn(N, device=DEV).to(dtype); r = torch.randn(M, N, device=DEV).to(dtype)
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    ref = x.float() @ w.float().t() + b.float() + r.float()
    line = f"{name:9s} {M}x{N}x{K}:"
    for cfg in base + WS:
        ctx = Ctx(DEV, dtype)
        y = ctx.gemm(x, w, bias=b, residual=r, cfg=cfg)
        err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
        us = graph_time(lambda c: c.gemm(x, w, bias=b, residual=r, out=out, cfg=cfg), dtype, n=20, reps=3) * 1e3
        line += f"  {cfg[0]}x{cfg[1]}/{cfg[2]} {us:6.1f}us" + ("" if err < 2e-2 else f" ERR {err:.1e}")
    print(line, flush=True)
CONVS = [("conv 64^2 640", 2, 64, 64, 640, 640, [(7128, 160, 1)]), ("conv 32^2 1280", 2, 32, 32, 1280, 1280, [(64, 128, 4), (7564, 160, 1)]), ("conv 32^2 640->1280", 2, 32, 32, 640, 1280, [(7564, 160, 1), (64, 128, 2)]),
         ("conv 32^2 2560->1280", 2, 32, 32, 2560, 1280, [(64, 128, 4)])]
for name, B, H, W, Cin, Cout, base in CONVS:
    x = torch.randn(B, H, W, Cin, device=DEV).to(dtype)
    w = (torch.randn(Cout, 3, 3, Cin, device=DEV) * (9 * Cin) ** -0.5).to(dtype)
    bias = torch.randn(Cout, device=DEV).to(dtype)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias.float(), padding=1).permute(0, 2, 3, 1)
    line = f"{name:20s}:"
    for cfg in base + WS + [(2464, 160, 2)]:
        ctx = Ctx(DEV, dtype)
        y = ctx.conv3x3(x, w.reshape(Cout, -1), bias, cfg=cfg)
        err = ((y.float().reshape(ref.shape) - ref).abs().max() / ref.abs().max()).item()
        us = graph_time(lambda c: c.conv3x3(x, w.reshape(Cout, -1), bias, cfg=cfg), dtype, n=10, reps=3) * 1e3
        line += f"  {cfg[0]}x{cfg[1]}/{cfg[2]} {us:6.1f}us" + ("" if err < 2e-2 else f" ERR {err:.1e}")
    print(line, flush=True)
