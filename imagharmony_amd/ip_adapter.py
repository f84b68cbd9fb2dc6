"""Adapter API of the reference (ip_adapter/ip_adapter.py) on the HIP hot path.

``install_ip_processors`` is ``IPAdapter.set_ip_adapter`` (ip_adapter.py:99-125): attn1 layers get
``AttnProcessor2_0``, attn2 layers get ``IPAttnProcessor2_0`` whose image-prompt branch is active only
under ``down_blocks.2.attentions.1`` (:117; the ``target_blocks`` argument is ignored upstream, :75).
"""
import torch

from .attention_processor import AttnProcessor2_0, CNAttnProcessor2_0, IPAttnProcessor2_0

ACTIVE_BLOCK = "down_blocks.2.attentions.1"


def install_ip_processors(unet, num_tokens=4, scale=1.0, device=None, dtype=torch.float16, init="default"):
    cfg = unet.config
    procs = {}
    for name in unet.attn_processors.keys():
        cross = None if name.endswith("attn1.processor") else cfg.cross_attention_dim
        if name.startswith("mid_block"):
            hidden = cfg.block_out_channels[-1]
        elif name.startswith("up_blocks"):
            hidden = list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
        else:
            hidden = cfg.block_out_channels[int(name[len("down_blocks.")])]
        if cross is None:
            procs[name] = AttnProcessor2_0()
        else:
            if init == "empty":      # skip the (slow) CPU kaiming init: weights are loaded / filled afterwards
                with torch.device("meta"):
                    p = IPAttnProcessor2_0(hidden, cross, scale=scale, num_tokens=num_tokens, skip=ACTIVE_BLOCK not in name)
                p = p.to_empty(device=device or "cpu").to(dtype)
            else:
                p = IPAttnProcessor2_0(hidden, cross, scale=scale, num_tokens=num_tokens, skip=ACTIVE_BLOCK not in name)
                p = p.to(device=device, dtype=dtype)
            procs[name] = p
    unet.set_attn_processor(procs)
    return procs


def set_scale(unet, scale):
    """IPAdapter.set_scale (ip_adapter.py:179-182) / StableDiffusionXLCustomPipeline.set_scale (custom_pipelines.py:17-20)"""
    for p in unet.attn_processors.values():
        if isinstance(p, IPAttnProcessor2_0):
            p.scale = scale


# --------------------------------------------------------------------------------------------
# Adapter classes (API surface of ip_adapter/ip_adapter.py:69-478 for the SDXL variants)
# --------------------------------------------------------------------------------------------
import os
from typing import List

from .modules import HarmonyAttention, ImageProjModel, MLPProjModel, Resampler  # noqa: E402,F401
from .utils import get_generator  # noqa: E402

IPAttnProcessor = IPAttnProcessor2_0


class IPAdapter:
    """``IPAdapter.__init__`` / ``set_ip_adapter`` / ``load_ip_adapter`` / ``get_image_embeds`` / ``set_scale``
    (ip_adapter.py:70-182).  Differences, all behaviour-compatible: the compute dtype is a parameter (the
    reference hard-codes fp16 in 8 places), ``number_class_crossattention`` may be None (ip_adapter.py:85
    crashes), the ``.safetensors`` branch works (ip_adapter.py:137-147 writes a missing key), and an already
    constructed CLIP vision model / pre-computed CLIP embeddings may be supplied (no network here)."""

    def __init__(self, sd_pipe, image_encoder_path, ip_ckpt, device, num_tokens=4, target_blocks=None,
                 number_class_crossattention=None, image_encoder=None, dtype=torch.float16, clip_embeddings_dim=1280,
                 clip_hidden_size=1280, clip_image_processor=None):
        self.device = device
        self.dtype = dtype
        self.image_encoder_path = image_encoder_path
        self.ip_ckpt = ip_ckpt
        self.num_tokens = num_tokens
        self.pipe = sd_pipe.to(self.device)
        self.set_ip_adapter()
        self.image_encoder = image_encoder
        self.clip_image_processor = clip_image_processor
        if image_encoder is not None and clip_image_processor is None:
            from transformers import CLIPImageProcessor                                     # ip_adapter.py:84
            self.clip_image_processor = CLIPImageProcessor()
        if image_encoder is None and image_encoder_path is not None:
            from transformers import CLIPImageProcessor, CLIPVisionModelWithProjection       # ip_adapter.py:81-84
            self.image_encoder = CLIPVisionModelWithProjection.from_pretrained(image_encoder_path).to(self.device, dtype=dtype)
            self.clip_image_processor = CLIPImageProcessor()
        self.clip_embeddings_dim = getattr(getattr(self.image_encoder, "config", None), "projection_dim", clip_embeddings_dim)
        self.clip_hidden_size = getattr(getattr(self.image_encoder, "config", None), "hidden_size", clip_hidden_size)
        self.number_class_crossattention = (number_class_crossattention.to(self.device, dtype=dtype)
                                            if number_class_crossattention is not None else None)
        self.image_proj_model = self.init_proj()
        if ip_ckpt is not None:
            self.load_ip_adapter()

    def _g(self, module):
        """the conditioning module as a replayed hipGraph (modules.Graphed): get_image_embeds runs the 0.8 ms replay, not an eager pass
        of ~45 launches; CPU tensors (the host-logic tests) go straight through"""
        from .modules import Graphed
        cache = self.__dict__.setdefault("_graphed", {})
        w = cache.get(id(module))
        if w is None or w.module is not module:
            w = cache[id(module)] = Graphed(module)
        return w

    def init_proj(self):                                                  # ip_adapter.py:91-97
        return ImageProjModel(cross_attention_dim=self.pipe.unet.config.cross_attention_dim,
                              clip_embeddings_dim=self.clip_embeddings_dim,
                              clip_extra_context_tokens=self.num_tokens).to(self.device, dtype=self.dtype)

    def set_ip_adapter(self):                                             # ip_adapter.py:99-133
        install_ip_processors(self.pipe.unet, num_tokens=self.num_tokens, device=self.device, dtype=self.dtype)
        cn = getattr(self.pipe, "controlnet", None)                       # :126-133: text-only cross-attention there
        if cn is not None:
            for c in (getattr(cn, "nets", None) or [cn]):
                c.set_attn_processor(CNAttnProcessor2_0(num_tokens=self.num_tokens))

    def load_ip_adapter(self, state_dict=None):                           # ip_adapter.py:135-154
        if state_dict is None:
            from .checkpoint import load_ip_adapter_file
            state_dict = load_ip_adapter_file(self.ip_ckpt)
        self.image_proj_model.load_state_dict(state_dict["image_proj"])
        if self.number_class_crossattention is not None:
            self.number_class_crossattention.load_state_dict(state_dict["composed_adapter"])
        ip_layers = torch.nn.ModuleList(self.pipe.unet.attn_processors.values())     # keys '<idx>.to_k_ip.weight'
        ip_layers.load_state_dict(state_dict["ip_adapter"])

    def _clip_embeds(self, pil_image, clip_image_embeds):
        if pil_image is not None:
            if self.image_encoder is None or self.clip_image_processor is None:
                raise NotImplementedError("the CLIP image encoder is the step before the hot path (SURVEY.md 8f-4): "
                                          "pass clip_image_embeds or construct with image_encoder_path")
            imgs = pil_image if isinstance(pil_image, (list, tuple)) else [pil_image]
            px = self.clip_image_processor(images=imgs, return_tensors="pt").pixel_values
            return self.image_encoder(px.to(self.device, dtype=self.dtype)).image_embeds
        return clip_image_embeds.to(self.device, dtype=self.dtype)

    @torch.inference_mode()
    def fused_clip_embeds(self, pil_image=None, clip_image_embeds=None, extra_prompt_embeds=None):
        """CLIP image embedding with the harmony-aware correction added: ``clip + HA(text, clip)`` (ip_adapter.py:163-173).
        Also the target the default PNS judge scores previews against (imagharmony_amd.pns.ClipPreferenceJudge)."""
        clip_image_embeds = self._clip_embeds(pil_image, clip_image_embeds)
        if extra_prompt_embeds is not None and self.number_class_crossattention is not None:
            extra = extra_prompt_embeds.to(self.device, self.dtype)
            clip_image_embeds = clip_image_embeds + self._g(self.number_class_crossattention)(extra, clip_image_embeds)   # :170-173
        return clip_image_embeds

    @torch.inference_mode()
    def get_image_embeds(self, pil_image=None, clip_image_embeds=None, extra_prompt_embeds=None):   # ip_adapter.py:158-177
        clip_image_embeds = self.fused_clip_embeds(pil_image, clip_image_embeds, extra_prompt_embeds)
        image_prompt_embeds = self._g(self.image_proj_model)(clip_image_embeds)
        uncond_image_prompt_embeds = self._g(self.image_proj_model)(torch.zeros_like(clip_image_embeds))
        return image_prompt_embeds, uncond_image_prompt_embeds

    def set_scale(self, scale):                                           # ip_adapter.py:179-182
        set_scale(self.pipe.unet, scale)

    def generate(self, pil_image=None, clip_image_embeds=None, prompt=None, negative_prompt=None, scale=1.0,
                 num_samples=4, seed=None, guidance_scale=7.5, num_inference_steps=30, prompt_embeds=None, **kwargs):
        """Base (SD-1.x style) generate, ip_adapter.py:184-247: the pipe's ``encode_prompt`` returns (cond, uncond)
        and the pipe takes no pooled embeddings.  ``prompt_embeds`` = that 2-tuple when no text encoder is attached."""
        self.set_scale(scale)
        if pil_image is not None:
            n = len(pil_image) if isinstance(pil_image, (list, tuple)) else 1
        else:
            n = clip_image_embeds.size(0)
        prompt = prompt if prompt is not None else "best quality, high quality"
        negative_prompt = negative_prompt if negative_prompt is not None else \
            "monochrome, lowres, bad anatomy, worst quality, low quality"
        if not isinstance(prompt, List):
            prompt = [prompt] * n
        if not isinstance(negative_prompt, List):
            negative_prompt = [negative_prompt] * n
        extra = {k: kwargs.pop(k) for k in ("uncond_clip_image_embeds",) if k in kwargs}   # IPAdapterPlus / Full, no CLIP attached
        ipe, uipe = self.get_image_embeds(pil_image=pil_image, clip_image_embeds=clip_image_embeds, **extra)
        bs, seq_len, _ = ipe.shape
        tile = lambda t: t.repeat(1, num_samples, 1).view(bs * num_samples, seq_len, -1)        # :216-220
        ipe, uipe = tile(ipe), tile(uipe)
        if prompt_embeds is None:
            prompt_embeds = self.pipe.encode_prompt(prompt, device=self.device, num_images_per_prompt=num_samples,
                                                    do_classifier_free_guidance=True, negative_prompt=negative_prompt)
        pe, ne = prompt_embeds[0], prompt_embeds[1]
        pe = torch.cat([pe.to(ipe.device, self.dtype), ipe], dim=1)                             # :230-231
        ne = torch.cat([ne.to(ipe.device, self.dtype), uipe], dim=1)
        self.generator = get_generator(seed, kwargs.pop("generator_device", "cpu"))
        return self.pipe(prompt_embeds=pe, negative_prompt_embeds=ne, guidance_scale=guidance_scale,
                         num_inference_steps=num_inference_steps, generator=self.generator, **kwargs).images

    # shared tail of the generate() variants
    def _run(self, image_prompt_embeds, uncond_image_prompt_embeds, prompt, negative_prompt, num_samples, seed,
             num_inference_steps, embeds, kwargs):
        bs, seq_len, _ = image_prompt_embeds.shape
        tile = lambda t: t.repeat(1, num_samples, 1).view(bs * num_samples, seq_len, -1)        # ip_adapter.py:302-306
        image_prompt_embeds, uncond_image_prompt_embeds = tile(image_prompt_embeds), tile(uncond_image_prompt_embeds)
        if embeds is None:
            embeds = self.pipe.encode_prompt(prompt, num_images_per_prompt=num_samples, do_classifier_free_guidance=True,
                                             negative_prompt=negative_prompt)
        pe, ne, ppe, npe = embeds
        dev = image_prompt_embeds.device
        pe = torch.cat([pe.to(dev, self.dtype), image_prompt_embeds], dim=1)                    # :321-322
        ne = torch.cat([ne.to(dev, self.dtype), uncond_image_prompt_embeds], dim=1)
        self.generator = get_generator(seed, kwargs.pop("generator_device", "cpu"))
        return self.pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=ppe,
                         negative_pooled_prompt_embeds=npe, num_inference_steps=num_inference_steps,
                         generator=self.generator, **kwargs).images


class IPAdapterXL(IPAdapter):
    """ip_adapter.py:249-340."""

    def __init__(self, sd_pipe, image_encoder_path, ip_ckpt, device, num_tokens=4, target_blocks=None, inference=False,
                 number_class_crossattention=None, **kw):
        self.inference = inference
        super().__init__(sd_pipe, image_encoder_path, ip_ckpt, device, num_tokens=num_tokens, target_blocks=target_blocks,
                         number_class_crossattention=number_class_crossattention, **kw)

    def generate(self, pil_image=None, prompt=None, negative_prompt=None, extra_text=None, scale=1.0, num_samples=4,
                 seed=None, num_inference_steps=30, clip_image_embeds=None, prompt_embeds=None, extra_prompt_embeds=None,
                 **kwargs):
        """Same arguments as the reference; additionally ``clip_image_embeds`` / ``prompt_embeds`` (a 4-tuple as
        returned by encode_prompt) / ``extra_prompt_embeds`` may be given when no encoders are attached."""
        self.set_scale(scale)
        # a stray number_class_crossattention= (test.py:38) travels on to the pipeline inside **kwargs, as upstream;
        # StableDiffusionXLCustomPipeline.__call__ here ignores unknown keywords like diffusers' stock pipeline does
        if pil_image is None and clip_image_embeds is not None:     # pre-computed CLIP embeddings carry the batch
            n = clip_image_embeds.size(0)
        else:
            n = 1 if not isinstance(pil_image, (list, tuple)) else len(pil_image)
        prompt = prompt if prompt is not None else "best quality, high quality"
        negative_prompt = negative_prompt if negative_prompt is not None else \
            "monochrome, lowres, bad anatomy, worst quality, low quality"
        if not isinstance(prompt, List):
            prompt = [prompt] * n
        if not isinstance(negative_prompt, List):
            negative_prompt = [negative_prompt] * n
        if extra_prompt_embeds is None and extra_text is not None:      # ip_adapter.py:285-297 (NameError upstream if None)
            extra_prompt_embeds = self.pipe.encode_prompt(extra_text, num_images_per_prompt=num_samples,
                                                          do_classifier_free_guidance=True,
                                                          negative_prompt=negative_prompt)[0]
        ipe, uipe = self.get_image_embeds(pil_image=pil_image, clip_image_embeds=clip_image_embeds,
                                          extra_prompt_embeds=extra_prompt_embeds)
        return self._run(ipe, uipe, prompt, negative_prompt, num_samples, seed, num_inference_steps, prompt_embeds, kwargs)


    @torch.no_grad()
    def generate_pns(self, seeds, pil_image=None, prompt=None, negative_prompt=None, extra_text=None, scale=1.0,
                     preview_steps=10, num_inference_steps=30, guidance_scale=5.0, scorer=None, batch=None,
                     clip_image_embeds=None, prompt_embeds=None, extra_prompt_embeds=None, height=None, width=None,
                     output_type="pil", **schedule_kw):
        """Preference-guided noise selection (README.md:27, assets/1.png) around ``generate``: every candidate seed gets
        a ``preview_steps`` denoise, a judge scores the previews, the best NOISE gets the full ``num_inference_steps``
        denoise.  Candidates are sharded over the ranks of an initialised torch.distributed group (one process per
        GPU; no per-step collective).  ``scorer`` (latents [S,4,h,w] -> [S]) defaults to the CLIP-space judge when a
        VAE and a CLIP vision model are attached, else to the latent statistic of ``pns.default_scorer``.
        ``batch`` = preview candidates stacked per UNet forward on a rank; None (default) = as many as the rank holds, up
        to 4 (BASELINE.json configs[4] runs 4 per GPU): a stacked forward costs 1.27x less per candidate than one at a
        time on MI355X (bench.py ``stacked_candidates`` / ``pns_two_stage``); the final denoise is batch 1 either way.
        Returns dict(images, best_seed, scores, latents)."""
        from . import pns
        self.set_scale(scale)
        pipe = self.pipe
        prompt = prompt if prompt is not None else "best quality, high quality"
        negative_prompt = negative_prompt if negative_prompt is not None else \
            "monochrome, lowres, bad anatomy, worst quality, low quality"
        if extra_prompt_embeds is None and extra_text is not None:
            extra_prompt_embeds = pipe.encode_prompt(extra_text, num_images_per_prompt=1, do_classifier_free_guidance=True,
                                                     negative_prompt=negative_prompt)[0]
        fused = self.fused_clip_embeds(pil_image, clip_image_embeds, extra_prompt_embeds)
        ipe = self._g(self.image_proj_model)(fused)
        uipe = self._g(self.image_proj_model)(torch.zeros_like(fused))
        if prompt_embeds is None:
            prompt_embeds = pipe.encode_prompt(prompt, num_images_per_prompt=1, do_classifier_free_guidance=True,
                                               negative_prompt=negative_prompt)
        pe, ne, ppe, npe = prompt_embeds
        pe = torch.cat([pe.to(ipe.device, self.dtype), ipe], dim=1)
        ne = torch.cat([ne.to(ipe.device, self.dtype), uipe], dim=1)
        height = height or pipe.default_sample_size * pipe.vae_scale_factor
        width = width or pipe.default_sample_size * pipe.vae_scale_factor
        if batch is None:
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
            batch = min(4, (len(list(seeds)) + world - 1) // world)
        S = max(1, int(batch))
        eng = pipe.engine
        rep = lambda t, n: t.repeat(*([n] + [1] * (t.ndim - 1)))
        if scorer is None:
            if pipe.vae is not None and self.image_encoder is not None:
                from .vae import decode_latents
                scorer = pns.ClipPreferenceJudge(lambda z: decode_latents(pipe.vae, z), self.image_encoder, fused)
            else:
                scorer = pns.default_scorer
        shape = (1, 4, height // 8, width // 8)

        def cond(n):
            eng.set_conditioning(rep(pe, n), rep(ne, n), rep(ppe.to(ipe.device), n), rep(npe.to(ipe.device), n), height, width,
                                 guidance_scale=guidance_scale)

        state = {"n": None}

        def preview(noise):
            if state["n"] != noise.shape[0]:
                cond(noise.shape[0]); state["n"] = noise.shape[0]
            eng.set_schedule(pipe.scheduler, preview_steps, **schedule_kw)
            return eng.denoise(noise).clone()

        def final(noise):
            if state["n"] != noise.shape[0]:
                cond(noise.shape[0]); state["n"] = noise.shape[0]
            eng.set_schedule(pipe.scheduler, num_inference_steps, **schedule_kw)
            return eng.denoise(noise).clone()

        r = pns.run_pns(preview, list(seeds), shape, scorer=scorer, device=self.device, final_fn=final, batch=S)
        out = r["latents"]
        if output_type != "latent":
            if pipe.vae is None and pipe.vae_decode is None:
                raise NotImplementedError("output_type=%r needs a VAE on the pipeline; pass output_type='latent'" % (output_type,))
            if pipe.vae is not None:
                from .vae import decode_latents, postprocess
                out = postprocess(decode_latents(pipe.vae, out), output_type)
            else:
                out = pipe.vae_decode(out)
        return dict(images=out, best_seed=r["best_seed"], scores=r["scores"], latents=r["latents"])


class IPAdapterPlus(IPAdapter):
    """ip_adapter.py:344-374: Resampler (12 heads, width = the UNet's cross-attention dim) over the penultimate CLIP
    hidden states; base ``generate``."""

    def init_proj(self):                                                  # ip_adapter.py:347-360
        cd = self.pipe.unet.config.cross_attention_dim
        return Resampler(dim=cd, depth=4, dim_head=64, heads=12, num_queries=self.num_tokens,
                         embedding_dim=self.clip_hidden_size, output_dim=cd, ff_mult=4).to(self.device, dtype=self.dtype)

    @torch.inference_mode()
    def get_image_embeds(self, pil_image=None, clip_image_embeds=None, uncond_clip_image_embeds=None):      # :362-374
        """clip_image_embeds here = penultimate hidden states [B, 257, hidden] (as upstream names them); without an
        attached CLIP model pass them together with ``uncond_clip_image_embeds`` (hidden states of a zero image)."""
        if pil_image is not None:
            if self.image_encoder is None or self.clip_image_processor is None:
                raise NotImplementedError("CLIP image encoder not attached: pass clip_image_embeds (hidden states)")
            imgs = pil_image if isinstance(pil_image, (list, tuple)) else [pil_image]
            px = self.clip_image_processor(images=imgs, return_tensors="pt").pixel_values.to(self.device, dtype=self.dtype)
            clip_image_embeds = self.image_encoder(px, output_hidden_states=True).hidden_states[-2]
            uncond_clip_image_embeds = self.image_encoder(torch.zeros_like(px), output_hidden_states=True).hidden_states[-2]
        if uncond_clip_image_embeds is None:
            raise NotImplementedError("pass uncond_clip_image_embeds (CLIP hidden states of an all-zero image)")
        c = clip_image_embeds.to(self.device, self.dtype)
        u = uncond_clip_image_embeds.to(self.device, self.dtype)
        return self._g(self.image_proj_model)(c), self._g(self.image_proj_model)(u)


class IPAdapterFull(IPAdapterPlus):
    """ip_adapter.py:377-386: every hidden-state token through the MLP projection."""

    def init_proj(self):
        return MLPProjModel(cross_attention_dim=self.pipe.unet.config.cross_attention_dim,
                            clip_embeddings_dim=self.clip_hidden_size).to(self.device, dtype=self.dtype)


class IPAdapterPlusXL(IPAdapter):
    """ip_adapter.py:389-478: Resampler over the penultimate CLIP hidden states."""

    def init_proj(self):                                                  # ip_adapter.py:391-403
        return Resampler(dim=1280, depth=4, dim_head=64, heads=20, num_queries=self.num_tokens,
                         embedding_dim=self.clip_hidden_size, output_dim=self.pipe.unet.config.cross_attention_dim,
                         ff_mult=4).to(self.device, dtype=self.dtype)

    @torch.inference_mode()
    def get_image_embeds(self, pil_image=None, clip_hidden_states=None, uncond_clip_hidden_states=None):   # :405-417
        if pil_image is not None:
            if self.image_encoder is None or self.clip_image_processor is None:
                raise NotImplementedError("CLIP image encoder not attached: pass clip_hidden_states")
            imgs = pil_image if isinstance(pil_image, (list, tuple)) else [pil_image]
            px = self.clip_image_processor(images=imgs, return_tensors="pt").pixel_values.to(self.device, dtype=self.dtype)
            clip_hidden_states = self.image_encoder(px, output_hidden_states=True).hidden_states[-2]
            uncond_clip_hidden_states = self.image_encoder(torch.zeros_like(px), output_hidden_states=True).hidden_states[-2]
        c = clip_hidden_states.to(self.device, self.dtype)
        u = uncond_clip_hidden_states.to(self.device, self.dtype)
        return self._g(self.image_proj_model)(c), self._g(self.image_proj_model)(u)

    def generate(self, pil_image=None, prompt=None, negative_prompt=None, scale=1.0, num_samples=4, seed=None,
                 num_inference_steps=30, clip_hidden_states=None, uncond_clip_hidden_states=None, prompt_embeds=None,
                 **kwargs):
        self.set_scale(scale)
        if pil_image is None and clip_hidden_states is not None:
            n = clip_hidden_states.size(0)
        else:
            n = 1 if not isinstance(pil_image, (list, tuple)) else len(pil_image)
        prompt = prompt if prompt is not None else "best quality, high quality"
        negative_prompt = negative_prompt if negative_prompt is not None else \
            "monochrome, lowres, bad anatomy, worst quality, low quality"
        if not isinstance(prompt, List):
            prompt = [prompt] * n
        if not isinstance(negative_prompt, List):
            negative_prompt = [negative_prompt] * n
        ipe, uipe = self.get_image_embeds(pil_image, clip_hidden_states, uncond_clip_hidden_states)
        return self._run(ipe, uipe, prompt, negative_prompt, num_samples, seed, num_inference_steps, prompt_embeds, kwargs)
