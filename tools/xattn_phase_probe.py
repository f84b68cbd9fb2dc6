"""Time line of the fused cross-attention launch per workgroup (xattn.hip built with -DXA_TIMING=1): entry / end of the to_q K loop / end of
the key loops / exit on the chip-wide 100 MHz counter, for the batch-2 (configs[1]) and batch-8 (configs[3]) calls, with and without the
image-prompt key set.     python tools/xattn_phase_probe.py build   (here) ;  gpurun -- python tools/xattn_phase_probe.py run"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "imagharmony_amd", "csrc"); OBJ = os.path.join(CSRC, "_obj"); TMP = os.path.join(ROOT, "tools", "tmp_libs")
# XA_ABL (second argument of `build`, env XA_ABL of `run`): the to_q K loop with parts removed (wrong results by design) -- 1 no MFMAs, 2 no fragment
# reads either, 4 no LDS-DMA inside the loop, 8 no barrier
ABL = int(sys.argv[2]) if len(sys.argv) > 2 else int(os.environ.get("XA_ABL", "0"))
LIBT = os.path.join(TMP, "libimh_xattn_timing.so" if not ABL else f"libimh_xattn_timing_abl{ABL}.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(TMP, exist_ok=True)
    o = os.path.join(TMP, f"xattn_timing{ABL}.o")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-ignored-attributes", "-Wno-unused-value", "-DXA_TIMING=1", f"-DXA_ABL={ABL}", "-I", CSRC,
                    "-c", os.path.join(CSRC, "xattn.hip"), "-o", o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBT, o] +
                   [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and not f.startswith("xattn")], check=True)
    os.remove(o); print("built", LIBT); sys.exit(0)
os.environ["IMH_LIB_PATH"] = LIBT
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
q = lambda t, f: float(t.kthvalue(max(1, int(f * t.numel())))[0])
for (name, B, H, Lq, T, mode) in [("cfg2 text only", 2, 20, 1024, 0, 1), ("cfg2 + 4 ip tokens", 2, 20, 1024, 4, 1), ("cfg4 batch 8 + 16 ip", 8, 20, 1024, 16, 1),
                                  ("C = 640, L = 4096", 2, 10, 4096, 0, 1),
                                  ("cfg4 batch 8 + 32 ip, WIDE", 8, 20, 1024, 32, 10), ("batch 8 text only, WIDE", 8, 20, 1024, 0, 10),
                                  ("C = 640, L = 4096, batch 8, WIDE", 8, 10, 4096, 0, 10), ("C = 640, L = 4096, batch 8", 8, 10, 4096, 0, 1),
                                  ] + ([] if ABL or not L.experimental() else [("cfg2 text only, 2 heads np2", 2, 20, 1024, 0, 3), ("cfg2 text only, 2 heads np4", 2, 20, 1024, 0, 4),
                                  ("cfg2 + 4 ip, 2 heads np4", 2, 20, 1024, 4, 4), ("cfg4 batch 8 + 16 ip, 2 heads np4", 8, 20, 1024, 16, 4)]):
    ctx.lib.imh_debug_set(3, mode)
    C_ = H * 64
    x = torch.randn(B * Lq, C_, device=DEV).to(dtype); wq = (torch.randn(C_, C_, device=DEV) * C_ ** -0.5).to(dtype)
    k = torch.randn(B * 128, C_, device=DEV).to(dtype); vt = torch.randn(C_, B * 128, device=DEV).to(dtype)
    k2 = torch.randn(B * 64, C_, device=DEV).to(dtype); vt2 = torch.randn(C_, B * 64, device=DEV).to(dtype)
    out = torch.empty(B * Lq, C_, device=DEV, dtype=dtype)
    rec = Ctx(DEV, dtype, record=True)
    kw = dict(k2=k2, vt2=vt2, Lk2=T, Lk2_pad=64, ldk2=C_, ldvt2=B * 64, scale2=1.0) if T else {}
    rec.cross_attention(x, wq, k, vt, out, B, H, Lq, 77, 128, C_, B * 128, 0.125, **kw)
    a = rec._ops[-1][1]
    items = (Lq // 128) * (H if mode in (1, 10) else H // 2) * B
    dbg = torch.zeros(8 * items + 128, dtype=torch.int64, device=DEV)      # (one record per WORKGROUP: up to two per item with half items)
    a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0
    for _ in range(3):
        L.check(ctx.lib.imh_cross_attention(C.byref(a), ctx.stream()), "xattn")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(ctx.lib.imh_cross_attention(C.byref(a), ctx.stream()), "xattn")
    e1.record(); torch.cuda.synchronize()
    d = dbg.cpu()[:8 * items].view(-1, 4)
    d = d[d[:, 0] != 0]
    t0 = int(d[:, 0].min())
    ent, pj, ky, ex = [(d[:, i] - t0).double() / 100.0 for i in range(4)]
    print(f"{name:24s} B={B} H={H} L={Lq} T={T}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us warm; {d.shape[0]} workgroups; entry (us after the first) median {q(ent, .5):.1f} / 90 % {q(ent, .9):.1f} / "
          f"max {float(ent.max()):.1f}; to_q K loop median {q(pj - ent, .5):.1f} / 90 % {q(pj - ent, .9):.1f} / max {float((pj - ent).max()):.1f} us; key loops median {q(ky - pj, .5):.1f} / max "
          f"{float((ky - pj).max()):.1f} us; store median {q(ex - ky, .5):.1f} us; last exit at {float(ex.max()):.1f} us", flush=True)
    if mode == 10:       # the wide form's K loop by segment (consumer wave 0 / producer wave 0 of workgroup 0, cycles per K tile)
        grid = 8 * ((items // (H // (H // 5)) * (H // 5) if False else (Lq // 128) * (H // 5) * B) + 7) // 8 * 8 // 8
        grid = 8 * (((Lq // 128) * (H // 5) * B + 7) // 8)
        t = dbg.cpu()[4 * grid: 4 * grid + 8].tolist()
        nt = max(t[3], 1)
        print(f"        per K tile -- consumer: reads+MFMAs (+ X loads) {t[0]/nt:.0f}, lgkmcnt {t[1]/nt:.0f}, barrier {t[2]/nt:.0f} = {sum(t[:3])/nt:.0f} cycles | producer: issue {t[4]/nt:.0f}, "
              f"vmcnt {t[5]/nt:.0f}, barrier {t[6]/nt:.0f} = {sum(t[4:7])/nt:.0f} cycles", flush=True)
