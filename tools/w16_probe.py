"""ff.net.0 on the sixteen-wave 256 x 320 tile (variant 26256 x 320, gemm_w16.hip) against the wave-specialised 256 x 160 (23256): bit-identity,
warm / cold time.   gpurun -- python tools/w16_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
TIMING = os.environ.get("W16_TIMING") == "1"      # tools/build_variant.sh w16_timing gemm_w16.hip -DW16_TIMING=1 ; IMH_LIB_PATH=tools/tmp_libs/lib_w16_timing.so
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from imagharmony_amd.attention_processor import fold_ln
DEV = "cuda:0"
junk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
for dtype in (torch.bfloat16, torch.float16):
    ctx = Ctx(DEV, dtype)
    for (M, N, K, flags) in [(2048, 10240, 1280, L.GF_LN_ROW | L.GF_GEGLU), (8192, 5120, 640, L.GF_LN_ROW | L.GF_GEGLU), (8192, 10240, 1280, L.GF_LN_ROW | L.GF_GEGLU),
                             (512, 640, 128, L.GF_LN_ROW), (256, 320, 64, L.GF_LN_ROW | L.GF_GEGLU)]:
        x = (torch.randn(M, K, device=DEV) * 1.3 + 0.4).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5)
        norm = torch.nn.LayerNorm(K)
        with torch.no_grad():
            norm.weight.copy_(1 + 0.2 * torch.randn(K)); norm.bias.copy_(0.3 * torch.randn(K))
        wg, s_, c_ = fold_ln(w, norm, ctx)
        st = ctx.row_stats(x)
        res = {}
        for cfg in ((23256, 160, 1), (26256, 320, 1), (26256, 320, 1, 'f8')):
            ctx.lib.imh_debug_set(9, 1 if len(cfg) == 4 else 0)      # key 9: 1 (default) = the 256 x 320 tile on eight fat waves, 0 = sixteen
            a, out, *_ = ctx.gemm(x, wg, flags=flags, ln=(s_, c_, 1e-5, st), cfg=cfg[:3], _args_only=True)
            for _ in range(3):
                L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm")
            e1.record(); torch.cuda.synchronize()
            warm = e0.elapsed_time(e1) / 20 * 1e3
            cold = []
            for _ in range(5):
                junk.fill_(1); torch.cuda.synchronize()
                e0.record(); L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm"); e1.record(); torch.cuda.synchronize()
                cold.append(e0.elapsed_time(e1) * 1e3)
            res['f8' if len(cfg) == 4 else cfg[0]] = (warm, sorted(cold)[2], out.clone())
            if TIMING and cfg[0] == 26256:
                q = lambda t, f: float(t.kthvalue(max(1, int(f * t.numel())))[0])
                dbg = torch.zeros(8 + 4 * 4096, dtype=torch.int64, device=DEV)
                a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0xfeed
                for lab in ("warm", "cold"):
                    if lab == "cold":
                        junk.fill_(1)
                    torch.cuda.synchronize(); dbg.zero_()
                    L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm"); torch.cuda.synchronize()
                    d = dbg.cpu()[8:].view(-1, 4); d = d[d[:, 0] != 0]
                    t0 = int(d[:, 0].min())
                    ent, l0, l1, ex = [(d[:, i] - t0).double() / 100.0 for i in range(4)]
                    print(f"    {lab}: {d.shape[0]} tiles; entry median {q(ent, .5):.1f} / max {float(ent.max()):.1f}; prologue median {q(l0 - ent, .5):.1f}; K loop median {q(l1 - l0, .5):.1f} / max "
                          f"{float((l1 - l0).max()):.1f}; epilogue median {q(ex - l1, .5):.1f} / max {float((ex - l1).max()):.1f}; last exit {float(ex.max()):.1f} us", flush=True)
                a.pf_ptr, a.pf_bytes = None, 0
        ctx.lib.imh_debug_set(9, 1)
        same = torch.equal(res[23256][2], res[26256][2])
        same8 = torch.equal(res[23256][2], res['f8'][2])
        d = (res[23256][2].float() - res[26256][2].float()).abs().max().item()
        fl = 2.0 * M * N * K
        print(f"{str(dtype)[6:]:9s} {M}x{N}x{K} flags={flags}: 256x160 {res[23256][0]:6.1f} us warm / {res[23256][1]:6.1f} cold | 256x320 sixteen waves {res[26256][0]:6.1f} us warm "
              f"({fl / res[26256][0] / 1e6:5.0f} TF) / {res[26256][1]:6.1f} cold | eight fat waves {res['f8'][0]:6.1f} us warm ({fl / res['f8'][0] / 1e6:5.0f} TF) / {res['f8'][1]:6.1f} cold | "
              f"bit-identical {same} / {same8} (max |d| {d:.2e})", flush=True)
