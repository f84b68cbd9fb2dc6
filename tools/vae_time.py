import sys, time, torch
sys.path.insert(0, '.')
from imagharmony_amd.vae import AutoencoderKL, decode_latents
DEV='cuda:0'
lat = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(0)).to(DEV) * 0.13025
for mdt in (torch.float16, torch.bfloat16):
    vae = AutoencoderKL().init_random_(1).to(DEV, mdt)
    for tiled in (False, True):
        vae.enable_tiling(tiled)
        decode_latents(vae, lat); torch.cuda.synchronize()
        t=time.perf_counter()
        for _ in range(3): decode_latents(vae, lat)
        torch.cuda.synchronize()
        print(mdt, vae.precision_for(), 'tiled' if tiled else 'untiled', (time.perf_counter()-t)/3*1e3, 'ms')
