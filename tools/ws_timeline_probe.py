"""Per-workgroup time line of wave-specialised GEMM launches with their REAL epilogues (gemm_ring.hip built with -DWS_TIMING=1):
entry -> K loop begin -> K loop end -> consumer wave 0's stores retired, on the chip-wide 100 MHz counter, for the UNet-batch-8 shapes
(M = 8192) next to the batch-2 ones.   python tools/ws_phase_probe.py build   (here) ;  gpurun -- python tools/ws_timeline_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
LIBT = os.path.join(ROOT, "tools", "tmp_libs", "libimh_ws_timing.so")
os.environ["IMH_LIB_PATH"] = LIBT
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from imagharmony_amd.attention_processor import fold_ln
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
q = lambda t, f: float(t.kthvalue(max(1, int(f * t.numel())))[0])
CASES = [  # name, M, N, K, cfg, kind: plain | res (bias + residual + LN statistics out) | geglu (folded LN + GEGLU) | qkv (folded LN)
    ("to_out b2", 2048, 1280, 1280, (2464, 160, 1), "res"), ("to_out b8", 8192, 1280, 1280, (23256, 160, 1), "res"),
    ("to_out b8 64x160", 8192, 1280, 1280, (2464, 160, 1), "res"), ("to_out b8 128x160", 8192, 1280, 1280, (24128, 160, 1), "res"),
    ("to_out b8 plain", 8192, 1280, 1280, (23256, 160, 1), "plain"),
    ("[Q|K|V] b2 (LN, plain store)", 2048, 3840, 1280, (23256, 128, 1), "qkv"), ("ff.out b2", 2048, 1280, 5120, (2464, 160, 1), "res"),
    ("ff.out b8", 8192, 1280, 5120, (23256, 160, 1), "res"), ("ff.net.0 b2", 2048, 10240, 1280, (23256, 160, 1), "geglu"),
    ("ff.net.0 b8", 8192, 10240, 1280, (23256, 160, 1), "geglu"), ("to_q-like b8 (LN)", 8192, 1280, 1280, (23256, 160, 1), "qkv"),
]
for (name, M, N, K, cfg, kind) in CASES:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
    tiles = 8 * 4096
    dbg = torch.zeros(8 + 4 * tiles, dtype=torch.int64, device=DEV)
    if kind in ("geglu", "qkv"):
        norm = torch.nn.LayerNorm(K)
        wg, s_, c_ = fold_ln(w.float(), norm, ctx)
        st = ctx.row_stats(x)
        a, _o, *_ = ctx.gemm(x, wg, flags=L.GF_LN_ROW | (L.GF_GEGLU if kind == "geglu" else 0), ln=(s_, c_, 1e-5, st), cfg=cfg, _args_only=True)
    elif kind == "res":
        res = torch.randn(M, N, device=DEV).to(dtype); bias = torch.randn(N, device=DEV).to(dtype)
        out = torch.empty(M, N, device=DEV, dtype=dtype)
        a, _o, *_ = ctx.gemm(x, w, out=out, bias=bias, residual=res, cfg=cfg, _args_only=True)
        wd = ctx.lib.imh_gemm_stats_slot_width(cfg[0], cfg[1])            # the LayerNorm statistics hand-over of the real launch
        if wd > 0 and N % wd == 0:
            stt = torch.empty(M, N // wd, 2, dtype=torch.float32, device=DEV)
            a.ln_stats_out, a.ln_slots_out = stt.data_ptr(), N // wd
    else:
        out = torch.empty(M, N, device=DEV, dtype=dtype)
        a, _o, *_ = ctx.gemm(x, w, out=out, cfg=cfg, _args_only=True)
    a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0xfeed
    for _ in range(3):
        dbg.zero_()
        L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm")
    e1.record(); torch.cuda.synchronize()
    us_warm = e0.elapsed_time(e1) / 20 * 1e3
    d_warm = dbg.cpu()[8:].view(-1, 4).clone()
    # one launch behind a cache-flushing copy (operands cold, as in the forward)
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    cold = []
    for _ in range(5):
        junk.fill_(1); torch.cuda.synchronize()
        e0.record(); L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm"); e1.record(); torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    del junk
    fl = 2.0 * M * N * K
    print(f"{name:20s} {M}x{N}x{K} {cfg} {kind}: {us_warm:.1f} us warm ({fl / us_warm / 1e6:.0f} TF), cold {sorted(cold)[2]:.1f} us", flush=True)
    for lab, d in (("warm", d_warm), ("cold", dbg.cpu()[8:].view(-1, 4))):
        d = d[d[:, 0] != 0]
        t0 = int(d[:, 0].min())
        ent, l0, l1, ex = [(d[:, i] - t0).double() / 100.0 for i in range(4)]
        pro, loop, epi = l0 - ent, l1 - l0, ex - l1
        print(f"    {lab}: {d.shape[0]} tiles; entry median {q(ent, .5):.1f} / 90 % {q(ent, .9):.1f} / max {float(ent.max()):.1f}; "
              f"prologue median {q(pro, .5):.1f} / max {float(pro.max()):.1f}; K loop median {q(loop, .5):.1f} / max {float(loop.max()):.1f}; "
              f"epilogue (K loop end -> stores retired) median {q(epi, .5):.1f} / 90 % {q(epi, .9):.1f} / max {float(epi.max()):.1f}; "
              f"last K loop ends {float(l1.max()):.1f}, last exit {float(ex.max()):.1f} us", flush=True)
