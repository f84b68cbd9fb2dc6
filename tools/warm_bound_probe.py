"""In-situ vs warm time of every launch of the recorded 1024^2 forward: for each op, (a) its time inside the forward
(HIP events around consecutive launches, every weight matrix cold) and (b) the time of its 3rd back-to-back repeat (weights and
activations warm in L2 / Infinity Cache).  The difference is what a perfect weight prefetch could recover."""
import sys, os, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from imagharmony_amd import lib as L
from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
from imagharmony_amd.schedulers import DDIMScheduler
DEV = torch.device("cuda:0")
L.load()
unet = bench.build_unet(DEV, torch.bfloat16, 4)
pe, ne, po, no = [t.to(DEV) for t in bench.synthetic_conditioning(4)]
pipe = StableDiffusionXLCustomPipeline(unet, scheduler=DDIMScheduler(), device=DEV, dtype=torch.bfloat16)
eng = pipe.engine
eng.set_conditioning(pe, ne, po, no, 1024, 1024, guidance_scale=5.0)
eng.set_schedule(pipe.scheduler, 30)
z = torch.randn(1, 4, 128, 128)
eng.denoise(z.to(DEV)); torch.cuda.synchronize()
rec = eng.plan
n = rec.lib.imh_plan_size(rec.plan)
insitu = rec.time_ops(); insitu = [min(a, b) for a, b in zip(insitu, rec.time_ops())]
s = rec.stream()
warm = []
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(n):
    for _ in range(2):
        L.check(rec.lib.imh_plan_run_range(rec.plan, i, i + 1, s), "run")
    ev[0].record()
    L.check(rec.lib.imh_plan_run_range(rec.plan, i, i + 1, s), "run")
    ev[1].record(); ev[1].synchronize()
    warm.append(ev[0].elapsed_time(ev[1]))
agg = collections.OrderedDict()
for t, a, b in zip(rec.tags, insitu, warm):
    k = t[2]
    d = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
    d[0] += 1; d[1] += a; d[2] += b; d[3] += t[3]
print(f"{n} ops; in-situ sum {sum(insitu):.2f} ms, warm sum {sum(warm):.2f} ms")
for k, (c, a, b, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} n={c:4d}  in-situ {a:7.3f} ms ({a / c * 1e3:7.1f} us)   warm {b:7.3f} ms ({b / c * 1e3:7.1f} us)   {fl / (a * 1e-3) / 1e12 if a else 0:7.1f} -> {fl / (b * 1e-3) / 1e12 if b else 0:7.1f} TF/s")
