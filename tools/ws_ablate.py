"""Timing ablations of the wave-specialised GEMM (gemm_ring.hip built with -DWS_ABL=<mask>): 1 no MFMAs, 2 no LDS-DMA,
4 no fragment reads; results are wrong by construction, only the time per launch is read."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "run":
    import torch
    from imagharmony_amd import lib as L
    from tools.gemm_bench import graph_time
    L.load()
    DEV = "cuda:0"; dtype = torch.bfloat16
    line = ""
    cfg = tuple(int(v) for v in os.environ.get("WS_CFG", "2464,160,1").split(","))
    shapes = [(2048, 1280, 1280), (2048, 1280, 5120)] if cfg[0] != 23256 else [(2048, 10240, 1280), (2048, 10240, 5120)]
    for (M, N, K) in shapes:
        x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
        out = torch.empty(M, N, device=DEV, dtype=dtype)
        for kw in ({}, dict(ldx=0, ldw=0)):
            us = graph_time(lambda c: c.gemm(x, w, out=out, cfg=cfg, **kw), dtype, n=20, reps=3) * 1e3
            line += f"  K={K}{' hot' if kw else ''} {us:6.1f}us"
    print(line, flush=True)
else:
    for mask in (0, 1, 2, 4, 5, 6, 7):
        lib = os.path.join(ROOT, "tools", "tmp_libs", f"lib_wsabl{mask}.so")
        if not os.path.exists(lib):
            continue
        env = dict(os.environ, IMH_LIB_PATH=lib)
        r = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True)
        print(f"WS_ABL={mask}:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
