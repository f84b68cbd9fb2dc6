"""Debug: folded-LN WS GEMM error map by (row block, column block)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from imagharmony_amd.attention_processor import fold_ln
DEV = "cuda:0"; dtype = torch.bfloat16
L.load(); ctx = Ctx(DEV, dtype)
for cfg in [(24128, 128, 1), (23256, 160, 1), (24128, 160, 1)]:
    for (M, N, K) in [(512, 640, 128), (512, 640, 320)]:
        g = torch.Generator("cpu").manual_seed(1)
        x = (torch.randn(M, K, generator=g) * 1.5 + 2.0).to(dtype).to(DEV)
        wf = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
        norm = torch.nn.LayerNorm(K, eps=1e-5).to(DEV)
        wg, s, c = fold_ln(wf, norm, ctx)
        y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5), cfg=cfg).float()
        ref = F.layer_norm(x.float(), (K,), norm.weight, norm.bias, 1e-5) @ wf.t()
        err = (y - ref).abs()
        rb = err.reshape(M // 32, 32, N).amax(dim=(1, 2))
        cb = err.reshape(M, N // 32, 32).amax(dim=(0, 2))
        print(cfg, (M, N, K), "max err", float(err.max()), "\n rows/32:", [round(float(v), 2) for v in rb], "\n cols/32:", [round(float(v), 2) for v in cb], flush=True)
