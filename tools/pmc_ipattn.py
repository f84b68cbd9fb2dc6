"""Workload for the SQ counter passes on the north-star call (rocprofv3 --pmc ... -- python tools/pmc_ipattn.py [cfg2|cfg4]): `reps`
IPAttnProcessor2_0 calls on an IP-active layer = fused [to_q + norm2 + text SDPA + image-prompt SDPA] launch (imh::xattn_kernel) + [to_out +
bias + residual] launch (imh::gemm_ws_kernel 64 x 160), back to back so that to_out reads what the fused launch just wrote, as in the
forward.  Nothing else from the GEMM family runs in this process: the kernel names in the counter database are exactly these two."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.attention_processor import IPAttnProcessor2_0
from imagharmony_amd.ctx import Ctx
from imagharmony_amd.unet import Attention, Norm
DEV = "cuda:0"; dtype = torch.bfloat16
B, T = (8, 16) if (len(sys.argv) > 1 and sys.argv[1] == "cfg4") else (2, 4)
Lq, C_, H, reps = 1024, 1280, 20, 10
g = torch.Generator(device="cpu").manual_seed(11)
attn, proc, norm = Attention(C_, H, cross_attention_dim=2048), IPAttnProcessor2_0(C_, 2048, scale=1.0, num_tokens=T), Norm(C_, 1e-5)
with torch.no_grad():
    for p in list(attn.parameters()) + list(proc.parameters()):
        p.copy_(torch.randn(p.shape, generator=g) * (p.shape[-1] ** -0.5 if p.ndim > 1 else 0.02))
    norm.weight.fill_(1.0); norm.bias.zero_()
attn, proc, norm = attn.to(DEV, dtype), proc.to(DEV, dtype), norm.to(DEV, dtype)
x = torch.randn(B * Lq, C_, generator=g).to(DEV, dtype)
ehs = torch.randn(B, 77 + T, 2048, generator=g).to(DEV, dtype)
pre = Ctx(DEV, dtype)
kv = proc.prepare_kv(pre, attn, ehs)
st = pre.row_stats(x)
torch.cuda.synchronize()
ctx = Ctx(DEV, dtype)
for _ in range(reps):
    y = proc.emit(ctx, attn, x, B, Lq, residual=x, kv=kv, ln=norm, ln_stats=st)
torch.cuda.synchronize()
