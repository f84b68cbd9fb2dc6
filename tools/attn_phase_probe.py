"""Cycles per 64-key tile of the attention key loop by phase (attention.hip built with -DATT_TIMING=1: cycle counter
at the phase boundaries, work item 0 / thread 0 writes the totals to args.pf_ptr):
[0] vmcnt wait + barrier  [1] LDS-DMA issue  [2] K reads + QK^T MFMA issue  [3] V^T read issue + MFMA drain + softmax
[4] PV MFMA issue.      python tools/attn_phase_probe.py build   (here) ;  gpurun -- python tools/attn_phase_probe.py run"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "imagharmony_amd", "csrc"); OBJ = os.path.join(CSRC, "_obj"); TMP = os.path.join(ROOT, "tools", "tmp_libs")
# `build wg` / `run wg`: -DATT_TIMING=2 -- every workgroup of the pipelined self-attention kernel stamps (entry, exit, HW_ID): the launch as a
# time line per workgroup and per CU (how much longer does a CU with two workgroups take?)
WG = len(sys.argv) > 2 and sys.argv[2] == "wg"
LIBT = os.path.join(TMP, "libimh_timing_wg.so" if WG else "libimh_timing.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(TMP, exist_ok=True)
    o = os.path.join(TMP, "attn_timing.o")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DATT_TIMING=2" if WG else "-DATT_TIMING=1", "-I", CSRC, "-c",
                    os.path.join(CSRC, "attention.hip"), "-o", o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBT, o] +
                   [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and not f.startswith("attention")], check=True)
    os.remove(o); print("built", LIBT); sys.exit(0)
os.environ["IMH_LIB_PATH"] = LIBT
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
if WG:
    for (B, H, Lq) in [(2, 20, 1024), (2, 10, 4096), (8, 20, 1024)]:
        C_ = H * 64
        qk = torch.randn(B * Lq, 2 * C_, device=DEV).to(dtype); vt = torch.randn(C_, B * Lq, device=DEV).to(dtype)
        o = torch.empty(B * Lq, C_, device=DEV, dtype=dtype)
        items = (Lq // 128) * H * B; grid = 8 * ((items + 7) // 8) * 2      # (up to per + 3 * split workgroups per XCD)
        dbg = torch.zeros(4 * grid + 8, dtype=torch.int64, device=DEV)
        a = L.AttnArgs()
        a.Q, a.K, a.Vt, a.O = qk.data_ptr(), qk[:, C_:].data_ptr(), vt.data_ptr(), o.data_ptr()
        a.B, a.H, a.Lq, a.Lk, a.Lk_pad = B, H, Lq, Lq, Lq
        a.ldq, a.ldk, a.ldvt, a.ldo = 2 * C_, 2 * C_, B * Lq, C_
        a.scale, a.dtype = 0.125, ctx.dt
        a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0
        for _ in range(3):
            L.check(ctx.lib.imh_attention(C.byref(a), ctx.stream()), "attention")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(ctx.lib.imh_attention(C.byref(a), ctx.stream()), "attention")
        e1.record(); torch.cuda.synchronize()
        d = dbg.cpu()[:4 * grid].view(-1, 4); d = d[d[:, 0] != 0]
        t0 = int(d[:, 0].min())
        ent = (d[:, 0] - t0).double() / 100.0; dur = (d[:, 1] - d[:, 0]).double() / 100.0; ex = (d[:, 1] - t0).double() / 100.0
        cu = ((d[:, 3] & 7) << 16) | ((d[:, 2] >> 8) & 0xff)     # (XCD by blockIdx & 7, SE / SH / CU bits of HW_ID)
        role = d[:, 3] >> 8                                      # 1 = key-quarter workgroup
        cus, cnt = torch.unique(cu, return_counts=True)
        per_cu = {int(c): int(n) for c, n in zip(cus, cnt)}
        load = torch.tensor([per_cu[int(c)] for c in cu])
        q = lambda t, f: float(t.kthvalue(max(1, int(f * t.numel())))[0])
        line = f"B={B} H={H} L={Lq}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us warm; {d.shape[0]} workgroups on {len(per_cu)} CUs; entry median {q(ent, .5):.1f} / max {float(ent.max()):.1f} us; last exit {float(ex.max()):.1f} us; workgroup duration by workgroups on its CU: "
        for n in sorted(set(per_cu.values())):
            for ro in (0, 1):
                m = (load == n) & (role == ro)
                if int(m.sum()):
                    line += f"[{n} per CU, {'key-quarter' if ro else 'whole'}: {int(m.sum())} wgs, median {q(dur[m], .5):.1f} / max {float(dur[m].max()):.1f} us] "
        print(line, flush=True)
    sys.exit(0)
names = ["wait+barrier", "dma issue", "K reads + QK MFMA", "V issue + softmax", "PV MFMA"]
for (B, H, Lq) in [(2, 20, 1024), (2, 10, 4096), (8, 10, 4096)]:
    C_ = H * 64
    qk = torch.randn(B * Lq, 2 * C_, device=DEV).to(dtype); vt = torch.randn(C_, B * Lq, device=DEV).to(dtype)
    o = torch.empty(B * Lq, C_, device=DEV, dtype=dtype); dbg = torch.zeros(8, dtype=torch.int64, device=DEV)
    a = L.AttnArgs()
    a.Q, a.K, a.Vt, a.O = qk.data_ptr(), qk[:, C_:].data_ptr(), vt.data_ptr(), o.data_ptr()
    a.B, a.H, a.Lq, a.Lk, a.Lk_pad = B, H, Lq, Lq, Lq
    a.ldq, a.ldk, a.ldvt, a.ldo = 2 * C_, 2 * C_, B * Lq, C_
    a.scale, a.dtype = 0.125, ctx.dt
    a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0
    for _ in range(3):
        L.check(ctx.lib.imh_attention(C.byref(a), ctx.stream()), "attention")
    torch.cuda.synchronize()
    d = dbg.cpu().tolist(); nt = max(d[5], 1)
    tot = sum(d[:5])
    print(f"B={B} H={H} L={Lq}: {nt} tiles, {tot/nt:.0f} cycles/tile: " +
          ", ".join(f"{n} {d[i]/nt:.0f}" for i, n in enumerate(names)), flush=True)
