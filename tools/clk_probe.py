"""Shader clock and socket power while (a) the 8192 x 5120 x 2560 calibration GEMM and (b) the recorded 1024^2 SDXL forward replay back to
back (rocm-smi sampled once a second from the host while a thread keeps the queue full): what the MFMA peak is at the clock the
part sustains under its power cap.   gpurun -- python tools/clk_probe.py > profiles/rNN_clocks_under_load.txt"""
import os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
from tools.sweep import DEV, build_unet, record

dtype = torch.bfloat16


def smi():
    t = subprocess.run("rocm-smi --showpower --showclocks 2>&1", shell=True, capture_output=True, text=True).stdout
    g = lambda pat: (re.search(pat, t) or [None, "?"])[1]
    return dict(sclk=g(r"sclk clock level: \S+ \((\d+)Mhz\)"), fclk=g(r"fclk clock level: \S+ \((\d+)Mhz\)"),
                mclk=g(r"mclk clock level: \S+ \((\d+)Mhz\)"), watts=g(r"Package Power \(W\): ([\d.]+)"))


def under_load(name, replay, seconds=5):
    stop = [False]

    def load():
        while not stop[0]:
            for _ in range(10):
                replay()
            torch.cuda.synchronize()
    th = threading.Thread(target=load); th.start()
    time.sleep(1.0)
    rows = []
    for _ in range(seconds):
        time.sleep(1.0)
        rows.append(smi())
    stop[0] = True; th.join()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10:.3f} ms per replay; " + "; ".join(f"sclk {r['sclk']} MHz {r['watts']} W" for r in rows), flush=True)


print("idle:", smi(), "max power:", subprocess.run("rocm-smi --showmaxpower 2>&1 | grep -i 'max graphics'", shell=True, capture_output=True, text=True).stdout.strip())
ctx = Ctx(DEV, dtype, record=True)
M, N, K = 8192, 5120, 2560
x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype); y = torch.empty(M, N, device=DEV, dtype=dtype)
for _ in range(50):
    ctx.gemm(x, w, out=y, cfg=(23256, 160, 1))
ctx.capture()
under_load("50 x gemm 8192x5120x2560 (23256 x 160)", ctx.replay)
u = build_unet(dtype)
rec, out, st = record(u, dtype, 128)
rec.capture()
under_load("SDXL forward 1024^2, CFG batch 2", rec.replay)
rec4, out4, st4 = record(u, dtype, 128, S=4)
rec4.capture()
under_load("SDXL forward 1024^2, UNet batch 8", rec4.replay)
