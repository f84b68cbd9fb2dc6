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
