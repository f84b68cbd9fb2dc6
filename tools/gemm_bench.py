"""GPU micro-benchmark + correctness check of GEMM / conv kernel variants on the SDXL shapes.
Usage: python tools/gemm_bench.py [--variants "128,128,1;256,128,1;..."]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagharmony_amd import lib as L          # noqa: E402
from imagharmony_amd.ctx import Ctx           # noqa: E402

DEV = "cuda:0"
SHAPES = [
    ("ff.geglu", 2048, 10240, 1280, None), ("ff.geglu", 8192, 5120, 640, None),
    ("ff.out", 2048, 1280, 5120, None), ("ff.out", 8192, 640, 2560, None),
    ("to_qk", 2048, 2560, 1280, None), ("to_q", 2048, 1280, 1280, None), ("to_qk", 8192, 1280, 640, None),
    ("to_out@64", 8192, 640, 640, None), ("shortcut", 2048, 1280, 2560, None), ("shortcut@64", 8192, 640, 1920, None),
    ("conv1@128", 32768, 320, 2880, (2, 128, 128, 320, 1, 0)), ("conv2@64", 8192, 640, 5760, (2, 64, 64, 640, 1, 0)),
    ("conv2@32", 2048, 1280, 11520, (2, 32, 32, 1280, 1, 0)), ("upsample", 8192, 1280, 11520, (2, 32, 32, 1280, 1, 1)),
    ("conv1@32cat", 2048, 1280, 23040, (2, 32, 32, 2560, 1, 0)),
    ("conv1@128cat", 32768, 320, 8640, (2, 128, 128, 960, 1, 0)), ("up@64->128", 32768, 640, 5760, (2, 64, 64, 640, 1, 1)),
]


def timeit(fn, n=10):
    """host-launched timing (Python + ctypes per call): only meaningful for kernels >> 20 us"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def graph_time(emit, dtype, n=20, reps=3):
    """GPU-side time per launch: n back-to-back launches recorded into a plan, captured into a hipGraph,
    replayed; includes the ~1.5 us inter-kernel boundary, excludes host launch overhead."""
    rec = Ctx(DEV, dtype, record=True)
    for _ in range(n):
        emit(rec)
    rec.capture()
    rec.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rec.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="128,128,1;64,128,1;64,64,1;256,128,1;256,256,1;3128,128,1;3064,64,1")
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
    variants = [tuple(int(v) for v in s.split(",")) for s in a.variants.split(";")]
    ctx = Ctx(DEV, dtype)
    print(torch.cuda.get_device_name(0))
    for name, M, N, K, geom in SHAPES:
        if geom:
            B, H, W, Cin, stride, up = geom
            x = torch.randn(B, H, W, Cin, device=DEV).to(dtype)
            w4 = (torch.randn(N, Cin, 3, 3, device=DEV) * K ** -0.5).to(dtype)
            w = w4.permute(0, 2, 3, 1).reshape(N, K).contiguous()
            xin = x.float().permute(0, 3, 1, 2)
            if up:
                xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
            ref = F.conv2d(xin, w4.float(), None, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(M, N)
            run = lambda cfg, c=None: (c or ctx).conv3x3(x, w, stride=stride, up=up, cfg=cfg).view(M, N)
        else:
            x = torch.randn(M, K, device=DEV).to(dtype)
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
            ref = x.float() @ w.float().t()
            run = lambda cfg, c=None: (c or ctx).gemm(x, w, cfg=cfg)
        line = f"{name:12s} M={M:6d} N={N:6d} K={K:6d}:"
        for cfg in variants:
            try:
                y = run(cfg)
                torch.cuda.synchronize()
                err = (y.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
                ms = graph_time(lambda c: run(cfg, c), dtype)
                tf = 2.0 * M * N * K / ms / 1e9
                line += f"  [{cfg[0]}x{cfg[1]}/{cfg[2]} {ms*1e3:6.1f}us {tf:6.0f}TF{'' if err < 0.02 else ' ERR=%.3f' % err}]"
            except Exception as ex:     # noqa: BLE001
                line += f"  [{cfg} FAIL {str(ex)[:60]}]"
        print(line, flush=True)


if __name__ == "__main__":
    main()
