"""Oracle denoise loop: ip_adapter/custom_pipelines.py:249-363 on pre-computed embeddings.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch

from .modules import IPAttnProcessor2_0, AttnProcessor2_0


def install_ip_processors(unet, num_tokens=4, scale=1.0):
    """ip_adapter/ip_adapter.py:99-125: attn1 -> AttnProcessor2_0, attn2 -> IPAttnProcessor2_0,
    IP branch active only under 'down_blocks.2.attentions.1' (:117)."""
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
            procs[name] = IPAttnProcessor2_0(hidden, cross, scale=scale, num_tokens=num_tokens,
                                            skip=("down_blocks.2.attentions.1" not in name))
    unet.set_attn_processor(procs)
    return procs


def set_scale(unet, scale):
    # custom_pipelines.py:17-20
    for p in unet.attn_processors.values():
        if isinstance(p, IPAttnProcessor2_0):
            p.scale = scale


@torch.no_grad()
def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """diffusers 0.30.0 pipeline_stable_diffusion_xl.rescale_noise_cfg (third-party, absent here; imported by
    custom_pipelines.py:6, called at :354): arXiv 2305.08891 section 3.4.  Per-sample unbiased std over (C, H, W)."""
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


def denoise(unet, scheduler, latents, prompt_embeds, negative_prompt_embeds, pooled, negative_pooled,
            height, width, num_inference_steps=30, guidance_scale=5.0,
            control_guidance_start=0.0, control_guidance_end=1.0, trace=None, guidance_rescale=0.0,
            original_size=None, crops_coords_top_left=(0, 0), target_size=None, denoising_end=None):
    """latents: [S,4,H/8,W/8] initial noise (already drawn on a CPU generator).
    prompt_embeds/negative_prompt_embeds: [S,77+T,2048]; pooled: [S,1280].
    Returns the final latents (output_type='latent')."""
    s = latents.shape[0]
    do_cfg = guidance_scale > 1.0                                              # :223
    scheduler.set_timesteps(num_inference_steps)                               # :250
    latents = latents * scheduler.init_noise_sigma                             # prepare_latents :255-265
    tid = torch.tensor([list(original_size or (height, width)) + list(crops_coords_top_left)
                        + list(target_size or (height, width))], dtype=prompt_embeds.dtype)   # :277-293
    ehs, text, ids = prompt_embeds, pooled, tid.repeat(s, 1)
    if do_cfg:                                                                 # :295-298
        ehs = torch.cat([negative_prompt_embeds, prompt_embeds], 0)
        text = torch.cat([negative_pooled, pooled], 0)
        ids = torch.cat([ids, ids], 0)
    cond_scale = next(p.scale for p in unet.attn_processors.values()
                      if isinstance(p, IPAttnProcessor2_0))                    # :319-322
    ts = scheduler.timesteps
    if denoising_end is not None and isinstance(denoising_end, float) and 0 < denoising_end < 1:      # :303-311
        cutoff = int(round(1000 - denoising_end * 1000))                       # scheduler.config.num_train_timesteps
        ts = ts[:len([t for t in ts if t >= cutoff])]
    for i, t in enumerate(ts):                                                 # :325
        if (i / len(ts) < control_guidance_start) or ((i + 1) / len(ts) > control_guidance_end):
            set_scale(unet, 0.0)                                               # :326-329
        else:
            set_scale(unet, cond_scale)
        x = torch.cat([latents] * 2) if do_cfg else latents                    # :332
        x = scheduler.scale_model_input(x, t)                                  # :334
        eps = unet(x, t, encoder_hidden_states=ehs,
                   added_cond_kwargs={"text_embeds": text, "time_ids": ids})[0]   # :337-345
        if do_cfg:                                                             # :348-350
            u, c = eps.chunk(2)
            eps = u + guidance_scale * (c - u)
            if guidance_rescale > 0.0:                                         # :351-354
                eps = rescale_noise_cfg(eps, c, guidance_rescale)
        latents = scheduler.step(eps, t, latents)[0]                           # :357
        if trace is not None:
            trace.append(latents.clone())
    set_scale(unet, cond_scale)
    return latents
