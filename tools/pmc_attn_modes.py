"""Self-attention at UNet batch 8 (every CU fully loaded: 5 workgroups per CU) under the pipelined kernel (mode 3) and the key-split kernel
(mode 5, -DIMH_EXPERIMENTAL), three launches each -- for rocprofv3 --pmc passes (tools/pmc_sq_summary.py):
    IMH_LIB_PATH=tools/tmp_libs/libimh_hip_experimental.so rocprofv3 --pmc ... -- python tools/pmc_attn_modes.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from imagharmony_amd import lib as L
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
for (B, H, Lq) in [(8, 20, 1024), (2, 10, 4096)]:
    C_ = H * 64
    qk = torch.randn(B * Lq, 2 * C_, device=DEV).to(dtype); vt = torch.randn(C_, B * Lq, device=DEV).to(dtype); o = torch.empty(B * Lq, C_, device=DEV, dtype=dtype)
    for mode in [3] + ([5] if L.experimental() else []):
        ctx.lib.imh_debug_set(4, mode)
        for _ in range(3):
            ctx.attention(qk[:, :C_], qk[:, C_:], vt, o, B, H, Lq, Lq, Lq, 2 * C_, 2 * C_, B * Lq, C_, 0.125)
        torch.cuda.synchronize()
ctx.lib.imh_debug_set(4, 0)
