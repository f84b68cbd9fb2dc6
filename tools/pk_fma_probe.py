"""Probe for the packed-fp32 hazard noted in imagharmony_amd/build.py: folded-LayerNorm GEMMs with cold ln_s / ln_c (build a variant WITH packed fp32 via tools/build_variant.sh and point IMH_LIB_PATH at it to reproduce).
Every iteration uses freshly allocated (cold) ln_s / ln_c vectors and flushes L2 / MALL first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from imagharmony_amd.attention_processor import fold_ln
DEV = torch.device("cuda:0")
L.load()
NIT = int(os.environ.get("LN_IT", "6"))

def rnd(*shape, dtype, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)

flush = torch.empty(768 << 20, dtype=torch.uint8, device=DEV)
tot_bad = 0
for dtype in (torch.bfloat16, torch.float16):
    for (M, N, K) in [(2048, 1280, 640), (2048, 1280, 1280), (8192, 1280, 640)]:
        for cfg in [(64, 64), (128, 64), (128, 128)]:
            ctx = Ctx(DEV, dtype)
            x = (rnd(M, K, dtype=dtype, seed=1) * 1.5 + 3.0).contiguous()
            w = rnd(N, K, dtype=torch.float32, seed=2, scale=K ** -0.5)
            norm = torch.nn.LayerNorm(K, eps=1e-5)
            with torch.no_grad():
                ref = (F.layer_norm(x.float().cpu(), (K,), norm.weight, norm.bias, 1e-5) @ w.cpu().t()).to(DEV)
            wg, s, c = fold_ln(w, norm, ctx)
            nbad, keep = [], []
            for it in range(NIT):
                si, ci = s.clone(), c.clone()
                keep += [si, ci]
                flush.fill_(it)
                torch.cuda.synchronize()
                y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(si, ci, 1e-5), cfg=(cfg[0], cfg[1], 1))
                torch.cuda.synchronize()
                nbad.append(int(((y.float() - ref).abs() > 0.1).sum()))
                ctx.free(y)
            tot_bad += sum(nbad)
            print(f"{str(dtype)[6:]} {M}x{N}x{K} cfg {cfg}: bad per cold run = {nbad}", flush=True)
print("TOTAL BAD", tot_bad)
