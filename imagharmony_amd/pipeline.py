"""Denoise pipeline with the call surface of the reference's ``StableDiffusionXLCustomPipeline``
(ip_adapter/custom_pipelines.py:16-394) for the part that is on the hot path: pre-computed prompt
embeddings in, latents out.  Text encoders, the VAE decode and PIL post-processing are the rows
SURVEY.md 8(f) marks "next"; they can be plugged in (``vae_decode`` / ``text_encoder`` callables) but are
not part of this library.
"""
from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Union

import torch

from .attention_processor import IPAttnProcessor
from .denoise import DenoiseEngine
from .schedulers import DDIMScheduler


@dataclass
class StableDiffusionXLPipelineOutput:
    images: Any


def randn_latents(shape, generator=None, dtype=torch.float32):
    """diffusers ``randn_tensor``: one generator, or a list with one generator per sample
    (ip_adapter/utils.py:83-93 builds exactly that for a list of seeds)."""
    if isinstance(generator, (list, tuple)):
        if len(generator) != shape[0]:
            raise ValueError(f"got {len(generator)} generators for a batch of {shape[0]}")
        return torch.cat([torch.randn((1,) + tuple(shape[1:]), generator=g, device=g.device, dtype=dtype).cpu()
                          for g in generator], 0)
    dev = generator.device if generator is not None else "cpu"
    return torch.randn(tuple(shape), generator=generator, device=dev, dtype=dtype).cpu()


class StableDiffusionXLCustomPipeline:
    vae_scale_factor = 8

    def __init__(self, unet, scheduler=None, device="cuda:0", dtype=torch.bfloat16, vae_decode: Optional[Callable] = None,
                 text_encoder: Optional[Callable] = None, use_graph=True, vae=None, watermark=None):
        """vae: an ``imagharmony_amd.vae.AutoencoderKL`` (HIP decode + post-processing for output_type 'pil' / 'np' /
        'pt'); ``vae_decode``: alternatively any callable latents -> images (its result is returned as is)."""
        self.unet = unet
        self.vae = vae
        self.watermark = watermark
        self.scheduler = scheduler or DDIMScheduler()
        self.device = torch.device(device)
        self.dtype = dtype
        self.vae_decode = vae_decode
        self.text_encoder = text_encoder
        self.default_sample_size = unet.config.sample_size
        self.engine = DenoiseEngine(unet, self.device, dtype, use_graph=use_graph)

    def to(self, device):
        self.device = torch.device(device)
        self.unet.to(self.device)
        self.engine = DenoiseEngine(self.unet, self.device, self.dtype, use_graph=self.engine.use_graph)
        return self

    def enable_vae_tiling(self):                                          # test.py:73
        if self.vae is None:
            raise NotImplementedError("no VAE attached (construct the pipeline with vae=AutoencoderKL)")
        self.vae.enable_tiling(True)

    def disable_vae_tiling(self):
        if self.vae is not None:
            self.vae.enable_tiling(False)

    def set_scale(self, scale):                                           # custom_pipelines.py:17-20
        for p in self.unet.attn_processors.values():
            if isinstance(p, IPAttnProcessor):
                p.scale = scale

    def encode_prompt(self, prompt, num_images_per_prompt=1, do_classifier_free_guidance=True, negative_prompt=None,
                      **kw):
        if self.text_encoder is None:
            raise NotImplementedError("text encoders are outside the hot path (SURVEY.md 8f-4): pass prompt_embeds, or "
                                      "construct the pipeline with text_encoder=callable")
        return self.text_encoder(prompt, num_images_per_prompt=num_images_per_prompt,
                                 do_classifier_free_guidance=do_classifier_free_guidance, negative_prompt=negative_prompt)

    @torch.no_grad()
    def __call__(self, prompt=None, height=None, width=None, num_inference_steps: int = 50, guidance_scale: float = 5.0,
                 negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 control_guidance_start: float = 0.0, control_guidance_end: float = 1.0, guidance_rescale: float = 0.0,
                 callback=None, callback_steps: int = 1, original_size=None, crops_coords_top_left=(0, 0),
                 target_size=None, denoising_end: Optional[float] = None, **kwargs):
        """``output_type`` defaults to "pil" as the reference does (custom_pipelines.py:42), so ``test.py:43``'s
        ``images[0].save(...)`` works; that needs a VAE (``vae=`` / ``vae_decode=``) -- without one the call fails
        BEFORE denoising with a clear error (pass output_type="latent" for latents).  The default scheduler is
        DDIM (BASELINE.json's metric; IP-Adapter convention) -- stock SDXL ships EulerDiscrete: pass
        ``scheduler=EulerDiscreteScheduler()`` for that."""
        if output_type != "latent" and self.vae is None and self.vae_decode is None:
            raise NotImplementedError("output_type=%r needs a VAE: construct the pipeline with "
                                      "vae=imagharmony_amd.vae.AutoencoderKL(...) or vae_decode=callable, or pass "
                                      "output_type='latent'" % (output_type,))
        if eta not in (0, 0.0):
            raise NotImplementedError("eta != 0 (stochastic DDIM) is not supported: the device-resident step is the "
                                      "deterministic x' = cx*x + ce*eps update (the reference runs eta = 0)")
        for k in ("negative_original_size", "negative_target_size", "prompt_2", "negative_prompt_2", "cross_attention_kwargs"):
            if kwargs.get(k) is not None:
                raise NotImplementedError(f"{k} is not supported on this path (custom_pipelines.py:23-56 accepts it for diffusers' sake)")
        height = height or self.default_sample_size * self.vae_scale_factor      # :189-190
        width = width or self.default_sample_size * self.vae_scale_factor
        if prompt_embeds is None:
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = \
                self.encode_prompt(prompt, num_images_per_prompt, guidance_scale > 1.0, negative_prompt)
        if pooled_prompt_embeds is None:
            raise ValueError("pooled_prompt_embeds must be passed together with prompt_embeds")     # check_inputs
        S = prompt_embeds.shape[0]
        eng = self.engine
        eng.set_conditioning(prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds,
                             height, width, guidance_scale, guidance_rescale=guidance_rescale, original_size=original_size,
                             crops_coords_top_left=crops_coords_top_left, target_size=target_size)
        eng.set_schedule(self.scheduler, num_inference_steps, control_guidance_start, control_guidance_end,
                         denoising_end=denoising_end)
        if latents is None:
            latents = randn_latents((S, 4, height // 8, width // 8), generator)       # prepare_latents :255-265
        out = eng.denoise(latents, callback=callback, callback_steps=callback_steps).clone()
        if output_type != "latent":                                       # custom_pipelines.py:365-386
            if self.vae is not None:
                from .vae import decode_latents, postprocess
                image = decode_latents(self.vae, out)
                if self.watermark is not None:
                    image = self.watermark.apply_watermark(image)
                out = postprocess(image, output_type)
            else:
                out = self.vae_decode(out)
        return StableDiffusionXLPipelineOutput(images=out) if return_dict else (out,)
