"""Adapter API of the reference (ip_adapter/ip_adapter.py) on the HIP hot path.

``install_ip_processors`` is ``IPAdapter.set_ip_adapter`` (ip_adapter.py:99-125): attn1 layers get
``AttnProcessor2_0``, attn2 layers get ``IPAttnProcessor2_0`` whose image-prompt branch is active only
under ``down_blocks.2.attentions.1`` (:117; the ``target_blocks`` argument is ignored upstream, :75).
"""
import torch

from .attention_processor import AttnProcessor2_0, IPAttnProcessor2_0

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
