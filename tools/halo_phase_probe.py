"""Cycles per (chunk, tap) step of the LDS-halo conv3x3 by segment (conv_halo.hip built with -DCH_TIMING=1): MFMA wave 0 of workgroup 0
[counted vmcnt] [s_barrier] [weight LDS-DMA issue] [fragment reads + MFMAs] and halo wave 0 [lgkmcnt] [s_barrier] [halo issue / transform].
    python tools/halo_phase_probe.py build   (here) ;  gpurun -- python tools/halo_phase_probe.py run"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "imagharmony_amd", "csrc"); OBJ = os.path.join(CSRC, "_obj"); TMP = os.path.join(ROOT, "tools", "tmp_libs")
LIBT = os.path.join(TMP, "libimh_halo_timing.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(TMP, exist_ok=True)
    o = os.path.join(TMP, "halo_timing.o")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-ignored-attributes", "-Wno-unused-value", "-DCH_TIMING=1", "-I", CSRC,
                    "-c", os.path.join(CSRC, "conv_halo.hip"), "-o", o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBT, o] +
                   [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and not f.startswith("conv_halo")], check=True)
    os.remove(o); print("built", LIBT); sys.exit(0)
os.environ["IMH_LIB_PATH"] = LIBT
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
G = 32
for (name, B, H, W, Cin, Cout, cfg, fused) in [
        ("conv2 @32 ks80 fused", 2, 32, 32, 1280, 1280, (7128, 80, 1), True), ("conv2 @32 ks80 plain", 2, 32, 32, 1280, 1280, (7128, 80, 1), False),
        ("conv2 @32 4x16x160 fused", 2, 32, 32, 1280, 1280, (7564, 160, 1), True),
        ("conv2 @64 8x16x160 fused", 2, 64, 64, 640, 640, (7128, 160, 1), True), ("conv2 @64 8x16x160 plain", 2, 64, 64, 640, 640, (7128, 160, 1), False),
        ("conv1 @128 16x16x160 fused", 2, 128, 128, 320, 320, (7256, 160, 1), True)]:
    x = torch.randn(B, H, W, Cin, device=DEV).to(dtype)
    w = (torch.randn(Cout, 9 * Cin, device=DEV) * (9 * Cin) ** -0.5).to(dtype)
    tab = torch.randn(B, Cin, 2, device=DEV, dtype=torch.float32) * 0.5 if fused else None
    dbg = torch.zeros(16, dtype=torch.int64, device=DEV)
    rec = Ctx(DEV, dtype, record=True)              # a recording context only to fill the argument struct
    rec.conv3x3(x, w, cfg=cfg, gn=(tab, True) if fused else None)
    a = rec._ops[-1][1]
    a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0
    for _ in range(3):
        L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "conv")
    e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().tolist()
    ns = max(d[4], 1)
    hs = max(d[11], 1)
    print(f"{name:28s} {cfg}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us warm; {ns} steps; per step -- MFMA wave: vmcnt {d[0]/ns:.0f}, barrier {d[1]/ns:.0f}, "
          f"issue {d[2]/ns:.0f}, reads+MFMAs {d[3]/ns:.0f} = {sum(d[:4])/ns:.0f} cycles | halo wave: lgkmcnt {d[8]/hs:.0f}, barrier {d[9]/hs:.0f}, "
          f"stage+transform {d[10]/hs:.0f} = {sum(d[8:11])/hs:.0f} cycles", flush=True)
