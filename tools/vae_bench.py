"""GPU: full-size SDXL VAE decode (random weights) of one 128x128 latent -> 1024x1024 image: time, finiteness,
tiled vs untiled.  python tools/vae_bench.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.vae import AutoencoderKL, decode_latents, postprocess
DEV = "cuda:0"
vae = AutoencoderKL().init_random_(1).to(DEV, torch.bfloat16)
lat = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(0)).to(DEV) * 0.13025
for tiled in (False, True):
    vae.enable_tiling(tiled)
    img = decode_latents(vae, lat); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        img = decode_latents(vae, lat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"tiled={tiled}: {dt*1e3:.1f} ms per 1024x1024 decode (eager, host-launched), shape {tuple(img.shape)}, finite {bool(torch.isfinite(img).all())}, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
t0 = time.perf_counter(); pil = postprocess(img, "pil"); print(f"postprocess -> PIL {pil[0].size}: {(time.perf_counter()-t0)*1e3:.1f} ms")
