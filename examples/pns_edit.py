"""The reference's test.py flow (test.py:28-104) on the MI355X path, with preference-guided noise selection over
several seeds.  Real checkpoints are optional -- without them every model is seeded random, which exercises the
whole path (PIL in -> CLIP-shaped embeddings -> HarmonyAttention + ImageProjModel -> 30-step denoise per candidate
-> PNS winner -> VAE tiled decode -> PIL out) but of course produces noise, not a picture.

    python examples/pns_edit.py --out out.png [--unet unet.safetensors] [--vae vae.safetensors] [--ip-ckpt ip_adapter.bin]
                                [--seeds 0 1 2 3] [--steps 30] [--preview-steps 10] [--size 1024]

Launch under torch.distributed.run with N ranks to shard the seeds over N GPUs (one process per GPU, RCCL).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagharmony_amd import pns                                           # noqa: E402
from imagharmony_amd.ip_adapter import IPAdapterXL                        # noqa: E402
from imagharmony_amd.modules import HarmonyAttention                      # noqa: E402
from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline      # noqa: E402
from imagharmony_amd.schedulers import DDIMScheduler                      # noqa: E402
from imagharmony_amd.unet import UNet2DConditionModel, UNetConfig         # noqa: E402
from imagharmony_amd.vae import AutoencoderKL, decode_latents, postprocess   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="out.png")
    ap.add_argument("--unet"); ap.add_argument("--vae"); ap.add_argument("--ip-ckpt")
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--preview-steps", type=int, default=10)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--guidance", type=float, default=5.0)
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl")
    dev, dtype = torch.device(f"cuda:{local}"), torch.bfloat16
    torch.cuda.set_device(dev)

    if a.unet:
        unet = UNet2DConditionModel.from_safetensors(a.unet, UNetConfig()).to(dev, dtype)
    else:
        with torch.device(dev):                                            # 2.6 B parameters: create them on the GPU
            unet = UNet2DConditionModel(UNetConfig())
        unet = unet.init_random_(1234).to(dtype)
    vae = AutoencoderKL.from_safetensors(a.vae, device=dev, dtype=dtype) if a.vae else AutoencoderKL().init_random_(1).to(dev, dtype)
    pns.broadcast_module_(unet)                                           # identical replicas on every rank
    pipe = StableDiffusionXLCustomPipeline(unet, scheduler=DDIMScheduler(), device=dev, dtype=dtype, vae=vae)
    pipe.enable_vae_tiling()                                              # test.py:73
    ha = HarmonyAttention(image_hidden_size=1280, text_context_dim=2048, inter_dim=2560, cross_heads=8, reshape_blocks=8,
                          cross_value_dim=64, scale=1.0, fusion_method="cross_attention")     # test.py:82-91
    ip = IPAdapterXL(pipe, None, a.ip_ckpt, dev, num_tokens=4, inference=True, number_class_crossattention=ha, dtype=dtype)

    # encoders are outside the path (no tokenizer vocabulary / CLIP weights offline): stand-in embeddings of the right shape
    g = torch.Generator().manual_seed(0)
    clip_embeds = torch.randn(1, 1280, generator=g)
    prompt = (torch.randn(1, 77, 2048, generator=g), torch.randn(1, 77, 2048, generator=g),
              torch.randn(1, 1280, generator=g), torch.randn(1, 1280, generator=g))
    extra = torch.randn(1, 77, 2048, generator=g)
    ipe, uipe = ip.get_image_embeds(clip_image_embeds=clip_embeds, extra_prompt_embeds=extra)
    ip.set_scale(a.scale)
    pe = torch.cat([prompt[0].to(dev, dtype), ipe], 1)
    ne = torch.cat([prompt[1].to(dev, dtype), uipe], 1)
    eng = pipe.engine
    eng.set_conditioning(pe, ne, prompt[2].to(dev, dtype), prompt[3].to(dev, dtype), a.size, a.size, guidance_scale=a.guidance)

    def denoise(steps):
        def f(noise):
            eng.set_schedule(pipe.scheduler, steps)
            return eng.denoise(noise).clone()
        return f

    r = pns.run_pns(denoise(a.preview_steps), a.seeds, (1, 4, a.size // 8, a.size // 8), device=dev, final_fn=denoise(a.steps))
    if int(os.environ.get("RANK", "0")) == 0:
        img = postprocess(decode_latents(vae, r["latents"]), "pil")[0]
        img.save(a.out)
        print(f"seeds {a.seeds} -> scores {[round(float(s), 4) for s in r['scores']]}; best seed {r['best_seed']} "
              f"(rank {r['owner']}); wrote {a.out} {img.size}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
