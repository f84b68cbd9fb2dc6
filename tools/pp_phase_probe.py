"""Cycles per K tile of the 256 x 320 ping-pong GEMM by segment (gemm_pp.hip built with -DPR_TIMING=1): for every phase
p = 0..3, [L issue: fragment reads + LDS-DMA issue] [L wait: lgkmcnt + barrier] [M: MFMA issue + vmcnt wait] [barrier];
wave 0 (first weight half) and wave 4 (second half, one barrier behind) of workgroup 0 write their totals to args.pf_ptr.
    python tools/pp_phase_probe.py build   (here) ;  gpurun -- python tools/pp_phase_probe.py run"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "imagharmony_amd", "csrc"); OBJ = os.path.join(CSRC, "_obj"); TMP = os.path.join(ROOT, "tools", "tmp_libs")
LIBT = os.path.join(TMP, "libimh_pp_timing.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(TMP, exist_ok=True)
    o = os.path.join(TMP, "pp_timing.o")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-ignored-attributes", "-DPR_TIMING=1", "-I", CSRC, "-c",
                    os.path.join(CSRC, "gemm_pp.hip"), "-o", o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBT, o] +
                   [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and not f.startswith("gemm_pp")], check=True)
    os.remove(o); print("built", LIBT); sys.exit(0)
os.environ["IMH_LIB_PATH"] = LIBT
import torch
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16
ctx = Ctx(DEV, dtype)
for (M, N, K) in [(2048, 10240, 1280), (256, 320, 1280), (2048, 10240, 5120)]:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
    out = torch.empty(M, N, device=DEV, dtype=dtype); dbg = torch.zeros(40, dtype=torch.int64, device=DEV)
    a, _o, *_ = ctx.gemm(x, w, out=out, cfg=(9256, 320, 1), _args_only=True)
    a.pf_ptr, a.pf_bytes = dbg.data_ptr(), 0
    for _ in range(3):
        L.check(ctx.lib.imh_gemm(C.byref(a), ctx.stream()), "gemm")
    torch.cuda.synchronize()
    d = dbg.cpu().tolist()
    for g in range(2):
        v = d[g * 17: g * 17 + 17]; nt = max(v[16], 1)
        tot = sum(v[:16])
        print(f"{M}x{N}x{K} wave {g * 4}: {nt} K tiles, {tot / nt:.0f} cycles / tile: " +
              " | ".join(f"p{ph}: L {v[ph*4]/nt:.0f}+{v[ph*4+1]/nt:.0f} M {v[ph*4+2]/nt:.0f}+{v[ph*4+3]/nt:.0f}" for ph in range(4)), flush=True)
