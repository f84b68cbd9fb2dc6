"""Race screen for the round-6 hand-synchronised kernels: many launches of gemm_f8 (256 x 320 ff.net.0 tile, eight fat waves) conv_hws
(wave-specialised LDS-halo conv) and the K-split conv's shared transform with fresh random operands, warm and behind cache-flushing traffic, each compared BIT FOR BIT with the older
kernel of the same variant code (gemm_w16: imh_debug_set(9, 0); conv_halo lock-step: imh_debug_set(5, 6)).  A barrier / counted-wait mistake
shows up as rare differing tiles that come and go with timing.   gpurun -- python tools/race_screen.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from imagharmony_amd.attention_processor import fold_ln
DEV = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
junk = torch.empty(384 << 20, dtype=torch.uint8, device=DEV)
bad = 0
for dtype in (torch.bfloat16, torch.float16):
    ctx = Ctx(DEV, dtype)
    for r in range(rounds):
        g = torch.Generator(device=DEV).manual_seed(1000 + r)
        # ---- gemm_f8 vs gemm_w16
        M, N, K = [(2048, 10240, 1280), (8192, 5120, 640), (512, 640, 64), (256, 320, 1280)][r % 4]
        x = (torch.randn(M, K, device=DEV, generator=g) * 1.3 + 0.4).to(dtype)
        w = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
        norm = torch.nn.LayerNorm(K)
        with torch.no_grad():
            norm.weight.copy_(1 + 0.2 * torch.randn(K)); norm.bias.copy_(0.3 * torch.randn(K))
        wg, s_, c_ = fold_ln(w, norm, ctx)
        st = ctx.row_stats(x)
        flags = L.GF_LN_ROW | (L.GF_GEGLU if r % 3 else 0)
        outs = []
        for form in (1, 0, 1, 1):
            ctx.lib.imh_debug_set(9, form)
            if len(outs) == 2:
                junk.fill_(r & 255)                 # the third launch runs cold, beside the flush's tail
            outs.append(ctx.gemm(x, wg, flags=flags, ln=(s_, c_, 1e-5, st), cfg=(26256, 320, 1)).clone())
        ctx.lib.imh_debug_set(9, 1)
        torch.cuda.synchronize()
        if not (torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3])):
            bad += 1; print(f"gemm_f8 MISMATCH round {r} {dtype} {M}x{N}x{K} flags={flags}", flush=True)
        # ---- conv_hws vs the lock-step kernels
        B, H, W, Cin, Cout, cfg = [(2, 64, 64, 320, 320, (7256, 160, 1)), (2, 32, 32, 640, 640, (7128, 160, 1)), (1, 48, 32, 960, 320, (7356, 160, 1)), (2, 16, 48, 1280, 640, (7128, 160, 1))][r % 4]
        xc = torch.randn(B, H, W, Cin, device=DEV, generator=g).to(dtype)
        wc = (torch.randn(Cout, 9 * Cin, device=DEV, generator=g) * (9 * Cin) ** -0.5).to(dtype)
        tab = (torch.randn(B, Cin, 2, device=DEV, generator=g) * 0.5).float().contiguous()
        res = torch.randn(B * H * W, Cout, device=DEV, generator=g).to(dtype)
        co = []
        for mode in (0, 6, 0, 0):
            ctx.lib.imh_debug_set(5, mode)
            if len(co) == 2:
                junk.fill_((r + 7) & 255)
            co.append(ctx.conv3x3(xc, wc, residual=res, cfg=cfg, gn=(tab, bool(r & 1))).clone())
        ctx.lib.imh_debug_set(5, 0)
        torch.cuda.synchronize()
        if not (torch.equal(co[0], co[1]) and torch.equal(co[0], co[2]) and torch.equal(co[0], co[3])):
            bad += 1; print(f"conv_hws MISMATCH round {r} {dtype} {(B, H, W, Cin, Cout)} {cfg}", flush=True)
        # ---- the K-split 32^2 form with its transform shared by sixteen waves vs by the service waves only (imh_debug_set(5, 8))
        B, H, W, Cin, Cout = [(2, 32, 32, 1280, 1280), (1, 16, 32, 2560, 640), (2, 16, 16, 640, 1280)][r % 3]
        xc = torch.randn(B, H, W, Cin, device=DEV, generator=g).to(dtype)
        wc = (torch.randn(Cout, 9 * Cin, device=DEV, generator=g) * (9 * Cin) ** -0.5).to(dtype)
        tab = (torch.randn(B, Cin, 2, device=DEV, generator=g) * 0.5).float().contiguous()
        co = []
        for mode in (0, 8, 0, 0):
            ctx.lib.imh_debug_set(5, mode)
            if len(co) == 2:
                junk.fill_((r + 3) & 255)
            co.append(ctx.conv3x3(xc, wc, cfg=(7128, 80, 1), gn=(tab, bool(r & 1))).clone())
        ctx.lib.imh_debug_set(5, 0)
        torch.cuda.synchronize()
        if not (torch.equal(co[0], co[1]) and torch.equal(co[0], co[2]) and torch.equal(co[0], co[3])):
            bad += 1; print(f"conv K-split MISMATCH round {r} {dtype} {(B, H, W, Cin, Cout)}", flush=True)
print(f"race screen: {2 * rounds} rounds x (gemm_f8, conv_hws, K-split conv: 4 launches each), mismatches: {bad}")
sys.exit(1 if bad else 0)
