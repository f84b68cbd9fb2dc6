"""Cycles per (chunk, tap) step of the wave-specialised LDS-halo conv3x3 by segment (conv_hws.hip built with -DHWS_TIMING=1): consumer wave 0 of
workgroup 0 [fragment reads + MFMAs] [own weight pieces issued / waited for + lgkmcnt(0)] [s_barrier], producer wave 0 [halo issue] [halo wait]
[transform + lgkmcnt] [s_barrier], and every wave's barrier wait (the wave with the smallest one is the step's critical path); next to the
launch time of the lock-step kernels (conv_halo.hip, imh_debug_set(5, 6)) on the same arguments.
    tools/build_variant.sh hws_timing conv_hws.hip -DHWS_TIMING=1   (here) ;  gpurun -- python tools/hws_phase_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["IMH_LIB_PATH"] = os.path.join(ROOT, "tools", "tmp_libs", "lib_hws_timing.so")
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
for (name, B, H, W, Cin, Cout, cfg, fused) in [
        ("conv @128 16x16x160 fused", 2, 128, 128, 320, 320, (7256, 160, 1), True), ("conv @128 16x16x160 plain", 2, 128, 128, 320, 320, (7256, 160, 1), False),
        ("conv @128 16x16x160 fused K=8640", 2, 128, 128, 960, 320, (7256, 160, 1), True),
        ("conv @64 8x16x160 fused", 2, 64, 64, 640, 640, (7128, 160, 1), True), ("conv @64 8x16x160 plain", 2, 64, 64, 640, 640, (7128, 160, 1), False),
        ("conv @64 8x16x160 fused K=17280", 2, 64, 64, 1920, 640, (7128, 160, 1), True)]:
    x = torch.randn(B, H, W, Cin, device=DEV).to(dtype)
    w = (torch.randn(Cout, 9 * Cin, device=DEV) * (9 * Cin) ** -0.5).to(dtype)
    tab = torch.randn(B, Cin, 2, device=DEV, dtype=torch.float32) * 0.5 if fused else None
    dbg = torch.zeros(32, dtype=torch.int64, device=DEV)
    rec = Ctx(DEV, dtype, record=True)              # a recording context only to fill the argument struct
    rec.conv3x3(x, w, cfg=cfg, gn=(tab, True) if fused else None)
    a = rec._ops[-1][1]
    a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0
    us = {}
    for mode in (0, 6):
        ctx.lib.imh_debug_set(5, mode)
        for _ in range(3):
            L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "conv")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "conv")
        e1.record(); torch.cuda.synchronize()
        us[mode] = e0.elapsed_time(e1) / 20 * 1e3
        if mode == 0:
            d = dbg.cpu().tolist()
    ctx.lib.imh_debug_set(5, 0)
    ns = max(d[4], 1)
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"{name:34s} {cfg}: {us[0]:.1f} us warm ({fl / us[0] / 1e6:.0f} TF/s; lock-step form {us[6]:.1f} us); {ns} steps; per step -- consumer: reads+MFMAs {d[0]/ns:.0f}, "
          f"weight pieces + lgkmcnt {d[1]/ns:.0f}, barrier {d[2]/ns:.0f} = {sum(d[:3])/ns:.0f} cycles | producer: halo issue {d[8]/ns:.0f}, halo wait {d[9]/ns:.0f}, "
          f"transform + lgkmcnt {d[10]/ns:.0f}, barrier {d[11]/ns:.0f} = {sum(d[8:12])/ns:.0f} cycles", flush=True)
    print("      barrier wait per step of every wave (consumers 0-7, producers 8-11): " + " ".join(f"{d[16 + w] / ns:.0f}" for w in range(12)), flush=True)
