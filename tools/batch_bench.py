"""Throughput of the denoise engine when S PNS candidates are batched into one UNet forward (UNet batch 2S)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
from imagharmony_amd.schedulers import DDIMScheduler
DEV = torch.device("cuda:0"); dtype = torch.bfloat16
unet = bench.build_unet(DEV, dtype, 4)
pe, ne, po, no = [t.to(DEV) for t in bench.synthetic_conditioning(4)]
for S in (1, 2, 4, 8):
    pipe = StableDiffusionXLCustomPipeline(unet, scheduler=DDIMScheduler(), device=DEV, dtype=dtype)
    eng = pipe.engine
    eng.set_conditioning(pe.repeat(S, 1, 1), ne.repeat(S, 1, 1), po.repeat(S, 1), no.repeat(S, 1), 1024, 1024, guidance_scale=5.0)
    eng.set_schedule(pipe.scheduler, 30)
    z = torch.randn(S, 4, 128, 128, device=DEV)
    eng.denoise(z); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.denoise(z); eng.denoise(z); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"S={S} candidates per forward: {dt*1e3:.0f} ms per batch -> {S/dt:.3f} images/sec ({dt/30*1e3:.1f} ms per UNet forward)", flush=True)
    del pipe, eng
