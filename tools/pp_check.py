"""gemm_pp (256 x 256 ping-pong GEMM, variant code 8256): correctness vs fp32, bitwise repeatability (race screen),
folded LayerNorm + GEGLU, and GPU-side timing against the 128 x 128 / ring variants on the GEGLU shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from imagharmony_amd.attention_processor import fold_ln
from tools.gemm_bench import graph_time
DEV = "cuda:0"
L.load()

def rnd(*shape, dtype, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)

ok_all = True
for PP in ((9256, 320, 1),):
  for dtype in (torch.bfloat16, torch.float16):
      ctx = Ctx(DEV, dtype)
      eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
      for (M, N, K) in [(128, 320, 64), (256, 640, 128), (256, 320, 192), (512, 768, 192), (300, 520, 192), (2048, 10240, 1280), (8192, 5120, 640), (2048, 2560, 1280)]:
          x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
          b, r = rnd(N, dtype=dtype, seed=3), rnd(M, N, dtype=dtype, seed=4)
          ref = x.float() @ w.float().t() + b.float() + r.float()
          ys = []
          for it in range(6):
              y = ctx.gemm(x, w, bias=b, residual=r, cfg=PP)
              torch.cuda.synchronize()
              ys.append(y.float().clone()); ctx.free(y)
          err = (ys[0] - ref).abs().max().item() / ref.abs().max().item()
          det = all(torch.equal(ys[0], t) for t in ys[1:])
          good = err < 4 * eps and det
          ok_all &= good
          print(f"{PP[0]} {str(dtype)[6:]:9s} plain {M}x{N}x{K}: rel max err {err:.2e} deterministic={det} {'OK' if good else 'FAIL'}", flush=True)
          if N % 32 == 0 and PP[0] in (8256, 9128, 9256):
              xl = (rnd(M, K, dtype=dtype, seed=5) * 1.5 + 2.0).contiguous()
              wf = rnd(N, K, dtype=torch.float32, seed=6, scale=K ** -0.5)
              norm = torch.nn.LayerNorm(K, eps=1e-5)
              with torch.no_grad():
                  norm.weight.copy_(1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3)))
                  norm.bias.copy_(0.3 * torch.randn(K, generator=torch.Generator().manual_seed(4)))
                  full = F.layer_norm(xl.float(), (K,), norm.weight.to(DEV), norm.bias.to(DEV), 1e-5) @ wf.t()
              gref = full[:, 0::2] * F.gelu(full[:, 1::2])
              wg, s, c = fold_ln(wf, norm, ctx)
              gs = []
              for it in range(4):
                  g = ctx.gemm(xl, wg, flags=L.GF_GEGLU | L.GF_LN_ROW, ln=(s.clone(), c.clone(), 1e-5), cfg=PP)
                  torch.cuda.synchronize()
                  gs.append(g.float().clone()); ctx.free(g)
              err = (gs[0] - gref).abs().max().item() / gref.abs().max().item()
              rms = ((gs[0] - gref).pow(2).mean().sqrt() / gref.pow(2).mean().sqrt()).item()
              det = all(torch.equal(gs[0], t) for t in gs[1:])
              good = err < 8 * eps and rms < 2 * eps and det
              ok_all &= good
              print(f"{PP[0]} {str(dtype)[6:]:9s} LN+GEGLU {M}x{N}x{K}: rel max err {err:.2e} rel-rms {rms:.2e} deterministic={det} {'OK' if good else 'FAIL'}", flush=True)
print("ALL OK" if ok_all else "FAILURES")
dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
for (M, N, K) in [(2048, 10240, 1280), (8192, 5120, 640), (2048, 2560, 1280), (8192, 1280, 640)]:
    xl = (rnd(M, K, dtype=dtype, seed=5) * 1.5 + 2.0).contiguous()
    wf = rnd(N, K, dtype=torch.float32, seed=6, scale=K ** -0.5)
    norm = torch.nn.LayerNorm(K, eps=1e-5)
    wg, s, c = fold_ln(wf, norm, ctx)
    line = f"timing {M}x{N}x{K}:"
    for cfg in [(128, 128, 1), (9128, 320, 1), (9256, 320, 1)]:
        ms = graph_time(lambda cx: cx.gemm(xl, wg, flags=L.GF_GEGLU | L.GF_LN_ROW, ln=(s, c, 1e-5), cfg=cfg), dtype)
        line += f"  LN+GEGLU {cfg[0]}x{cfg[1]} {ms * 1e3:6.1f}us {2.0 * M * N * K / ms / 1e9:5.0f}TF"
    for cfg in [(128, 128, 1), (8256, 256, 1), (9128, 320, 1), (5258, 320, 1), (9256, 320, 1)]:
        ms = graph_time(lambda cx: cx.gemm(xl, wg, cfg=cfg), dtype)
        line += f"  plain {cfg[0]}x{cfg[1]} {ms * 1e3:6.1f}us {2.0 * M * N * K / ms / 1e9:5.0f}TF"
    print(line, flush=True)
