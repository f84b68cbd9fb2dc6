import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
from imagharmony_amd.schedulers import DDIMScheduler
DEV = torch.device("cuda:0"); dtype = torch.bfloat16
unet = bench.build_unet(DEV, dtype, 4)
pe, ne, po, no = [t.to(DEV) for t in bench.synthetic_conditioning(4)]
pipe = StableDiffusionXLCustomPipeline(unet, scheduler=DDIMScheduler(), device=DEV, dtype=dtype)
eng = pipe.engine
eng.set_conditioning(pe, ne, po, no, 1024, 1024, guidance_scale=5.0); eng.set_schedule(pipe.scheduler, 30)
z = torch.randn(1, 4, 128, 128, device=DEV)
eng.denoise(z); torch.cuda.synchronize()
for trial in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.plan.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"10 graph replays: host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms -> {100*(t2-t0):.2f} ms/replay", flush=True)
# same plan without graph (C++ launches)
eng.plan.captured = False
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng.plan.run()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"10 plan runs (no graph): host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms", flush=True)
