"""Warm / cold per-launch time of the fused cross-attention's one-head form (imh_debug_set(3, 1)) against the wide form (3, 10) on the SDXL
call shapes (round 6).   gpurun -- python tools/xattn_wide_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"
junk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
for dtype in (torch.bfloat16, torch.float16):
    ctx = Ctx(DEV, dtype)
    for (name, B, H, Lq, T) in [("cfg4 b8 L1024 T32", 8, 20, 1024, 32), ("cfg3 b8 L1024 T16", 8, 20, 1024, 16), ("b8 L1024 text only", 8, 20, 1024, 0),
                                ("b8 C640 L4096 T16", 8, 10, 4096, 16), ("b8 C640 L4096 text", 8, 10, 4096, 0), ("b4 L1024 T16", 4, 20, 1024, 16),
                                ("cfg2 b2 L1024 T4", 2, 20, 1024, 4), ("b2 C640 L4096 text", 2, 10, 4096, 0)]:
        C_ = H * 64
        x = torch.randn(B * Lq, C_, device=DEV).to(dtype); wq = (torch.randn(C_, C_, device=DEV) * C_ ** -0.5).to(dtype)
        k = torch.randn(B * 128, C_, device=DEV).to(dtype); vt = torch.randn(C_, B * 128, device=DEV).to(dtype)
        k2 = torch.randn(B * 64, C_, device=DEV).to(dtype); vt2 = torch.randn(C_, B * 64, device=DEV).to(dtype)
        res = {}
        for mode in (1, 10):
            ctx.lib.imh_debug_set(3, mode)
            out = torch.empty(B * Lq, C_, device=DEV, dtype=dtype)
            rec = Ctx(DEV, dtype, record=True)
            kw = dict(k2=k2, vt2=vt2, Lk2=T, Lk2_pad=64, ldk2=C_, ldvt2=B * 64, scale2=1.0) if T else {}
            rec.cross_attention(x, wq, k, vt, out, B, H, Lq, 77, 128, C_, B * 128, 0.125, **kw)
            a = rec._ops[-1][1]
            for _ in range(3):
                L.check(ctx.lib.imh_cross_attention(C.byref(a), ctx.stream()), "xattn")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                L.check(ctx.lib.imh_cross_attention(C.byref(a), ctx.stream()), "xattn")
            e1.record(); torch.cuda.synchronize()
            warm = e0.elapsed_time(e1) / 20 * 1e3
            cold = []
            for _ in range(5):
                junk.fill_(1); torch.cuda.synchronize()
                e0.record(); L.check(ctx.lib.imh_cross_attention(C.byref(a), ctx.stream()), "xattn"); e1.record(); torch.cuda.synchronize()
                cold.append(e0.elapsed_time(e1) * 1e3)
            res[mode] = (warm, sorted(cold)[2], out.clone())
        ctx.lib.imh_debug_set(3, 0)
        fl = 2.0 * B * Lq * C_ * C_ + 4.0 * B * H * Lq * (77 + T) * 64
        same = torch.equal(res[1][2], res[10][2])
        print(f"{str(dtype)[6:]:9s} {name:22s}: one-head {res[1][0]:6.1f} us warm / {res[1][1]:6.1f} cold | wide {res[10][0]:6.1f} us warm ({fl / res[10][0] / 1e6:5.0f} TF) / {res[10][1]:6.1f} cold | bit-identical {same}", flush=True)
