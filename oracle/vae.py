"""CPU restatement of diffusers 0.30.0 ``AutoencoderKL`` for the SDXL VAE -- TEST INFRASTRUCTURE ONLY.

The reference decodes the final latents with the pipeline's VAE (ip_adapter/custom_pipelines.py:365-379, fp32
upcast because the SDXL VAE overflows in fp16; test.py:73 enables tiling) and post-processes with
``VaeImageProcessor.postprocess`` (:386).  diffusers is a third-party dependency that is neither vendored under
/root/reference nor installed here (requirements.txt:25, diffusers==0.30.0), so this file restates its published
architecture and algorithms; **parity unpinned** -- the structural pin is the exact parameter count of the SDXL
VAE (83,653,863) and the state-dict key schema (tests/test_oracle_vae.py).

Restated pieces: Encoder / Decoder (ResnetBlock2D without time embedding, single-head mid-block attention with
GroupNorm and residual, nearest-x2 Upsample2D + conv, asymmetric-pad stride-2 Downsample2D), ``decode`` with
``post_quant_conv``, ``tiled_decode`` with ``blend_v`` / ``blend_h``, and ``postprocess`` (denormalise, NHWC, numpy / PIL).
"""
import math
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    sample_size: int = 1024
    scaling_factor: float = 0.13025          # SDXL (SD 1.x: 0.18215)
    force_upcast: bool = True
    tile_overlap_factor: float = 0.25


def sdxl_vae_config():
    return VAEConfig()


def tiny_vae_config():
    """reduced width for GPU parity tests (channels stay multiples of 64 for the implicit-GEMM conv kernel)"""
    return VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1, sample_size=256)


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D with temb_channels=None: GN -> SiLU -> conv -> GN -> SiLU -> (dropout 0) -> conv, + shortcut"""

    def __init__(self, cin, cout, groups, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class VAEAttention(nn.Module):
    """diffusers Attention as the VAE mid block uses it: heads = 1 (attention_head_dim = channels), group_norm over
    the spatial tokens, linear projections WITH bias, residual connection, rescale_output_factor 1"""

    def __init__(self, channels, groups, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Identity()])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x.view(b, c, h * w)).transpose(1, 2)          # [B, L, C]
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(c), dim=-1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(b, c, h, w) + x


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))                          # diffusers pads right/bottom when padding=0


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    def __init__(self, cin, cout, n, groups, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(n)])
        if down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        if up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        for m in getattr(self, "downsamplers", []):
            x = m(x)
        for m in getattr(self, "upsamplers", []):
            x = m(x)
        return x


class _Mid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.attentions = nn.ModuleList([VAEAttention(ch, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, groups), ResnetBlock2D(ch, ch, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = ch[0]
        for i, co in enumerate(ch):
            self.down_blocks.append(_Block(c, co, cfg.layers_per_block, cfg.norm_num_groups, down=i < len(ch) - 1))
            c = co
        self.mid_block = _Mid(ch[-1], cfg.norm_num_groups)
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch = tuple(reversed(cfg.block_out_channels))
        self.conv_in = nn.Conv2d(cfg.latent_channels, ch[0], 3, padding=1)
        self.mid_block = _Mid(ch[0], cfg.norm_num_groups)
        self.up_blocks = nn.ModuleList()
        c = ch[0]
        for i, co in enumerate(ch):
            self.up_blocks.append(_Block(c, co, cfg.layers_per_block + 1, cfg.norm_num_groups, up=i < len(ch) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: VAEConfig = None):
        super().__init__()
        self.config = cfg or VAEConfig()
        c = self.config
        self.encoder = Encoder(c)
        self.decoder = Decoder(c)
        self.quant_conv = nn.Conv2d(2 * c.latent_channels, 2 * c.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(c.latent_channels, c.latent_channels, 1)
        self.use_tiling = False
        self.tile_sample_min_size = c.sample_size if c.sample_size < 512 else 512
        self.tile_latent_min_size = int(self.tile_sample_min_size / (2 ** (len(c.block_out_channels) - 1)))
        self.tile_overlap_factor = c.tile_overlap_factor

    def enable_tiling(self, on=True):
        self.use_tiling = on

    # -- diffusers AutoencoderKL.blend_v / blend_h --
    @staticmethod
    def blend_v(a, b, extent):
        extent = min(a.shape[2], b.shape[2], extent)
        for y in range(extent):
            b[:, :, y, :] = a[:, :, -extent + y, :] * (1 - y / extent) + b[:, :, y, :] * (y / extent)
        return b

    @staticmethod
    def blend_h(a, b, extent):
        extent = min(a.shape[3], b.shape[3], extent)
        for x in range(extent):
            b[:, :, :, x] = a[:, :, :, -extent + x] * (1 - x / extent) + b[:, :, :, x] * (x / extent)
        return b

    def tiled_decode(self, z):
        """diffusers AutoencoderKL.tiled_decode: overlapping latent tiles, linear blend of the overlaps"""
        overlap = int(self.tile_latent_min_size * (1 - self.tile_overlap_factor))
        extent = int(self.tile_sample_min_size * self.tile_overlap_factor)
        limit = self.tile_sample_min_size - extent
        rows = []
        for i in range(0, z.shape[2], overlap):
            row = []
            for j in range(0, z.shape[3], overlap):
                t = z[:, :, i:i + self.tile_latent_min_size, j:j + self.tile_latent_min_size]
                row.append(self.decoder(self.post_quant_conv(t)))
            rows.append(row)
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, t in enumerate(row):
                if i > 0:
                    t = self.blend_v(rows[i - 1][j], t, extent)
                if j > 0:
                    t = self.blend_h(row[j - 1], t, extent)
                out.append(t[:, :, :limit, :limit])
            out_rows.append(torch.cat(out, dim=3))
        return torch.cat(out_rows, dim=2)

    @torch.no_grad()
    def decode(self, z):
        if self.use_tiling and (z.shape[-1] > self.tile_latent_min_size or z.shape[-2] > self.tile_latent_min_size):
            return self.tiled_decode(z)
        return self.decoder(self.post_quant_conv(z))


def decode_latents(vae: AutoencoderKL, latents):
    """custom_pipelines.py:365-379: image = vae.decode(latents / scaling_factor) in fp32"""
    return vae.decode(latents.float() / vae.config.scaling_factor)


def postprocess(image, output_type="pil"):
    """VaeImageProcessor.postprocess with do_normalize=True (custom_pipelines.py:386): [-1,1] -> [0,1], clamp;
    'pt' -> [B,3,H,W]; 'np' -> float32 [B,H,W,3]; 'pil' -> list of RGB images (uint8 = round(x*255))"""
    x = (image / 2 + 0.5).clamp(0, 1)
    if output_type == "pt":
        return x
    arr = x.cpu().permute(0, 2, 3, 1).float().numpy()
    if output_type == "np":
        return arr
    from PIL import Image
    return [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
