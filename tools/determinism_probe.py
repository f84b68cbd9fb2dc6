"""GPU tool: which knob makes the full-size denoise non-deterministic?  Runs the 2-step 1024^2 denoise twice per configuration
(fresh pipeline each time) and reports bitwise equality of the two results.
Knobs: attn (imh_debug_set 4), xattn (3), ln_stats (unet.LN_STATS_HANDOVER), dual_ws."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from imagharmony_amd import lib as L, unet as U, attention_processor as AP
from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
from imagharmony_amd.schedulers import DDIMScheduler

DEV = torch.device("cuda:0")
lib = L.load()
unet = bench.build_unet(DEV, torch.bfloat16, 4)
pe, ne, po, no = [t.to(DEV) for t in bench.synthetic_conditioning(4)]


def run(pipe, seed=3, steps=2):
    z = torch.randn(1, 4, 128, 128, generator=torch.Generator("cpu").manual_seed(seed))
    return pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=po, negative_pooled_prompt_embeds=no, height=1024, width=1024,
                num_inference_steps=steps, guidance_scale=5.0, latents=z, output_type="latent").images.clone()


CONFIGS = [("default", {}), ("attn1", dict(attn=1)), ("ln_stats_off", dict(ln_stats=False)), ("attn1_ln_off", dict(attn=1, ln_stats=False)),
           ("attn2_ln_off", dict(attn=2, ln_stats=False))]
out = {}
for name, c in CONFIGS:
    lib.imh_debug_set(4, int(c.get("attn", 0)))
    lib.imh_debug_set(3, int(c.get("xattn", 0)))
    U.LN_STATS_HANDOVER = bool(c.get("ln_stats", True))
    pipe = StableDiffusionXLCustomPipeline(unet, scheduler=DDIMScheduler(), device=DEV, dtype=torch.bfloat16)
    a = run(pipe)
    eq = []
    for i in range(4):
        b = run(pipe)
        eq.append(bool(torch.equal(a, b)))
    out[name] = dict(equal=eq, maxdiff=float((a - b).abs().max()))
    print(name, out[name], flush=True)
    del pipe
    torch.cuda.empty_cache()
print(json.dumps(out))
