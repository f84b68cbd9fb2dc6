"""GPU probe: cost of the LayerNorm-fold pieces in isolation (hipGraph replay timing).
producer (+row statistics), finalize, consumer (+folded LN epilogue), vs the LayerNorm kernel they replace."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from tools.gemm_bench import graph_time
DEV = "cuda:0"; dtype = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(0)
def rnd(*s, scale=1.0, dt=dtype): return (torch.randn(*s, device=DEV, generator=g) * scale).to(dt)
for (M, C_) in [(2048, 1280), (8192, 640)]:
    x = rnd(M, C_); w = rnd(C_, C_, scale=C_ ** -0.5); b = rnd(C_); res = rnd(M, C_)
    out = torch.empty(M, C_, device=DEV, dtype=dtype); part = torch.empty(((C_ + 31) // 32) * M * 2, device=DEV)
    cfg = Ctx(DEV, dtype)._config(M, C_, C_, 0, 0)
    def prod(c, rs):
        r = c.gemm(x, w, bias=b, residual=res, out=out, cfg=cfg, rowstats=rs)
        if rs: c.free(r[1])
    t0 = graph_time(lambda c: prod(c, False), dtype); t1 = graph_time(lambda c: prod(c, True), dtype)
    c0 = Ctx(DEV, dtype); _, pt = c0.gemm(x, w, bias=b, residual=res, cfg=cfg, rowstats=True)
    tf = graph_time(lambda c: c.free(c.layernorm_stats(out, 1e-5, partials=pt)), dtype)
    ts = graph_time(lambda c: c.free(c.layernorm_stats(out, 1e-5)), dtype)
    gam, bet = rnd(C_), rnd(C_)
    tl = graph_time(lambda c: c.free(c.layernorm(out, gam, bet, 1e-5)), dtype)
    stat = c0.layernorm_stats(out, 1e-5); s = rnd(2 * C_, dt=torch.float32); cc = rnd(2 * C_, dt=torch.float32)
    w2 = rnd(2 * C_, C_, scale=C_ ** -0.5); o2 = torch.empty(M, 2 * C_, device=DEV, dtype=dtype)
    cfg2 = c0._config(M, 2 * C_, C_, 0, 0)
    tc0 = graph_time(lambda c: c.gemm(x, w2, out=o2, cfg=cfg2), dtype)
    tc1 = graph_time(lambda c: c.gemm(x, w2, out=o2, cfg=cfg2, flags=L.GF_LN_ROW, ln=(stat, s, cc)), dtype)
    print(f"M={M} C={C_} cfg={cfg}: producer {t0*1e3:.1f} -> {t1*1e3:.1f} us (+stats) | finalize {tf*1e3:.1f} us, two-pass stats {ts*1e3:.1f} us, "
          f"LayerNorm {tl*1e3:.1f} us | consumer N=2C cfg={cfg2} {tc0*1e3:.1f} -> {tc1*1e3:.1f} us (+folded LN)", flush=True)
