"""Where the LN + GEGLU launch (ff.net.0) spends its extra time over a plain GEMM of the same shape: plain / GEGLU only /
folded LayerNorm only / both, per tile variant, GPU-side timing (same box, one process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from imagharmony_amd.attention_processor import fold_ln
from tools.gemm_bench import graph_time
DEV = "cuda:0"; dtype = torch.bfloat16
L.load()
ctx = Ctx(DEV, dtype)
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 10240, 1280)
x = (torch.randn(M, K, device=DEV) * 1.5 + 2.0).to(dtype)
wf = torch.randn(N, K, device=DEV) * K ** -0.5
norm = torch.nn.LayerNorm(K, eps=1e-5)
wg, s, c = fold_ln(wf, norm, ctx)
bias = torch.randn(N, device=DEV).to(dtype)
CFGS = [(128, 128, 1), (9128, 320, 1), (23256, 160, 1), (24128, 160, 1), (24128, 128, 1)]
if os.environ.get('GEGLU_CFG'):
    CFGS = [tuple(int(v) for v in os.environ['GEGLU_CFG'].split(','))]
for cfg in CFGS:
    line = f"{cfg[0]}x{cfg[1]}:"
    for name, kw in [("plain", dict()), ("bias", dict(bias=bias)), ("GEGLU", dict(flags=L.GF_GEGLU, bias=bias)),
                     ("LN", dict(flags=L.GF_LN_ROW, ln=(s, c, 1e-5), bias=bias)),
                     ("LN+GEGLU", dict(flags=L.GF_LN_ROW | L.GF_GEGLU, ln=(s, c, 1e-5), bias=bias))]:
        ms = min(graph_time(lambda cx: cx.gemm(x, wg, cfg=cfg, **kw), dtype) for _ in range(2))
        line += f"  {name} {ms * 1e3:6.1f}us"
    print(line, flush=True)
