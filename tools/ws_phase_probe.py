"""Cycles per K tile of the wave-specialised GEMM by segment (gemm_ring.hip built with -DWS_TIMING=1): consumer wave 0 of workgroup 0
[fragment reads + MFMAs] [lgkmcnt(0)] [s_barrier] and producer wave 0 [LDS-DMA issue] [vmcnt] [s_barrier], totals over the K loop
written to args.pf_ptr.   python tools/ws_phase_probe.py build   (here) ;  gpurun -- python tools/ws_phase_probe.py run"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "imagharmony_amd", "csrc"); OBJ = os.path.join(CSRC, "_obj"); TMP = os.path.join(ROOT, "tools", "tmp_libs")
LIBT = os.path.join(TMP, "libimh_ws_timing.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(TMP, exist_ok=True)
    o = os.path.join(TMP, "ws_timing.o")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-ignored-attributes", "-Wno-unused-value", "-DWS_TIMING=1", "-I", CSRC,
                    "-c", os.path.join(CSRC, "gemm_ring.hip"), "-o", o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBT, o] +
                   [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and not f.startswith("gemm_ring")], check=True)
    os.remove(o); print("built", LIBT); sys.exit(0)
os.environ["IMH_LIB_PATH"] = LIBT
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
for (name, M, N, K, cfg) in [("ff.out", 2048, 1280, 5120, (2464, 160, 1)), ("to_out", 2048, 1280, 1280, (2464, 160, 1)),
                             ("geglu (no LN)", 2048, 10240, 1280, (23256, 160, 1)), ("big", 8192, 5120, 2560, (23256, 160, 1)),
                             ("ff.out @64", 8192, 640, 2560, (24128, 160, 1))]:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
    out = torch.empty(M, N, device=DEV, dtype=dtype); dbg = torch.zeros(16, dtype=torch.int64, device=DEV)
    a, _o, *_ = ctx.gemm(x, w, out=out, cfg=cfg, _args_only=True)
    a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0
    for _ in range(3):
        L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm")
    e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().tolist()
    nt = max(d[3], 1)
    print(f"{name:14s} {M}x{N}x{K} {cfg}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us warm; {nt} K tiles; per tile -- consumer: reads+MFMAs {d[0]/nt:.0f}, "
          f"lgkmcnt {d[1]/nt:.0f}, barrier {d[2]/nt:.0f} = {sum(d[:3])/nt:.0f} cycles | producer: issue {d[4]/nt:.0f}, vmcnt {d[5]/nt:.0f}, "
          f"barrier {d[6]/nt:.0f} = {sum(d[4:7])/nt:.0f} cycles", flush=True)

# ---- per-workgroup time line (100 MHz counter): entry -> K loop begin -> K loop end, grouped by the CU the workgroup ran on
import collections
from imagharmony_amd.attention_processor import fold_ln
for (name, M, N, K, cfg, ln, persist) in [("geglu (no LN)", 2048, 10240, 1280, (23256, 160, 1), False, 1), ("ff.out", 2048, 1280, 5120, (2464, 160, 1), False, 1),
                                          ("big", 8192, 5120, 2560, (23256, 160, 1), False, 1),
                                          ("ff.net.0", 2048, 10240, 1280, (23256, 160, 1), True, 1)]:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
    tiles = 8 * 4096
    dbg = torch.zeros(8 + 4 * tiles, dtype=torch.int64, device=DEV)
    if ln:
        norm = torch.nn.LayerNorm(K)
        wg, s_, c_ = fold_ln(w.float(), norm, ctx)
        st = ctx.row_stats(x)
        a, _o, *_ = ctx.gemm(x, wg, flags=L.GF_LN_ROW | L.GF_GEGLU, ln=(s_, c_, 1e-5, st), cfg=cfg, _args_only=True)
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
    d = dbg.cpu()[8:].view(-1, 4)
    d = d[d[:, 0] != 0]
    t0 = int(d[:, 0].min())
    ent, l0, l1 = (d[:, 0] - t0).double() / 100.0, (d[:, 1] - t0).double() / 100.0, (d[:, 2] - t0).double() / 100.0      # us
    pro, loop = l0 - ent, l1 - l0
    by_cu = collections.defaultdict(list)
    for i in range(d.shape[0]):
        by_cu[int(d[i, 3])].append((float(ent[i]), float(l0[i]), float(l1[i])))
    gaps = []
    for v in by_cu.values():
        v.sort()
        gaps += [b[0] - a_[2] for a_, b in zip(v, v[1:])]
    q = lambda t, f: float(t.kthvalue(max(1, int(f * t.numel())))[0])
    print(f"{name:14s} {M}x{N}x{K} {cfg}: {us_warm:.1f} us warm; {d.shape[0]} tiles; entry (us after the first) median {q(ent, .5):.1f} / 90 % {q(ent, .9):.1f} / max {float(ent.max()):.1f}; "
          f"prologue median {q(pro, .5):.1f} / 90 % {q(pro, .9):.1f} / max {float(pro.max()):.1f} us; K loop median {q(loop, .5):.1f} / max {float(loop.max()):.1f} us; "
          f"last K loop ends at {float(l1.max()):.1f} us" + (f"; next entry on the same CU {sorted(gaps)[len(gaps)//2]:.1f} us (median) after the previous K loop's end" if gaps else ""), flush=True)
