"""GPU probe: is one CFG batch-2 forward faster as two concurrent batch-1 forwards on two HIP streams?
(guidance_scale = 1 makes the engine run batch 1; two forked engines in flight = the two CFG halves)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from imagharmony_amd import pns
from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
from imagharmony_amd.schedulers import DDIMScheduler
DEV = torch.device("cuda:0"); dtype = torch.bfloat16
unet = bench.build_unet(DEV, dtype, 4)
pe, ne, po, no = [t.to(DEV) for t in bench.synthetic_conditioning(4)]
pipe = StableDiffusionXLCustomPipeline(unet, scheduler=DDIMScheduler(), device=DEV, dtype=dtype)
z = pns.seed_latents(0, (1, 4, 128, 128)).to(DEV)
def timed(fn, n=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
eng = pipe.engine
eng.set_conditioning(pe, ne, po, no, 1024, 1024, guidance_scale=5.0); eng.set_schedule(pipe.scheduler, 30)
t_b2 = timed(lambda: eng.denoise(z))
eng.set_conditioning(pe, ne, po, no, 1024, 1024, guidance_scale=1.0); eng.set_schedule(pipe.scheduler, 30)
t_b1 = timed(lambda: eng.denoise(z))
e2 = eng.fork(); streams = [torch.cuda.Stream(DEV) for _ in range(2)]
def pair():
    cur = torch.cuda.current_stream(DEV)
    for e, s in zip((eng, e2), streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s): e.denoise(z)
    for s in streams: cur.wait_stream(s)
t_pair = timed(pair)
print(f"30 steps: one batch-2 forward stream {t_b2*1e3:.0f} ms ({t_b2/30*1e3:.2f} ms/step) | one batch-1 stream {t_b1*1e3:.0f} ms "
      f"({t_b1/30*1e3:.2f}) | two batch-1 streams concurrently {t_pair*1e3:.0f} ms ({t_pair/30*1e3:.2f} ms/step)")
