"""Oracle restatement of diffusers==0.30.0 ``UNet2DConditionModel`` (SDXL-base config).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

diffusers is the third-party dependency that holds this arithmetic
(reference requirements.txt:25; call sites ip_adapter/custom_pipelines.py:338-345,
train.py:310).  It is not vendored in /root/reference and not installed, so this
file restates its published architecture.  What pins it:
the exact SDXL parameter count (2,567,463,684), the state-dict key schema, the
140-entry ``attn_processors`` dict (70 attn1 + 70 attn2, 10 of them under
``down_blocks.2.attentions.1``) -- see tests/test_oracle_unet.py.  **Numerical
parity with diffusers itself is unpinned** (no diffusers to execute).

Module / parameter names follow the diffusers state-dict schema so real SDXL
weights would load unchanged.
"""
import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import AttnProcessor2_0


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    sample_size: int = 128
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    layers_per_block: int = 2
    # per down block; block 0 is a plain DownBlock2D (no attention)
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20)      # = number of heads (head_dim 64)
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    norm_num_groups: int = 32
    norm_eps: float = 1e-5

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4

    @property
    def pooled_dim(self):
        return self.projection_class_embeddings_input_dim - 6 * self.addition_time_embed_dim

    def get(self, k, d=None):
        return getattr(self, k, d)


def sdxl_config():
    return UNetConfig()


def tiny_config():
    """Reduced-width config with the same topology (used by fast parity tests)."""
    return UNetConfig(block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2),
                      attention_head_dim=(1, 2, 4), cross_attention_dim=256, addition_time_embed_dim=64,
                      projection_class_embeddings_input_dim=128 + 6 * 64, sample_size=32)


def timestep_embedding(t, dim):
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``; fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    a = t[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class Attention(nn.Module):
    """diffusers ``Attention(query_dim, heads, dim_head=64, bias=False, out_bias=True)`` with
    the processor protocol the reference plugs into (ip_adapter/ip_adapter.py:99-125)."""

    def __init__(self, query_dim, heads, dim_head=64, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])
        self.processor = AttnProcessor2_0()

    def get_processor(self):
        return self.processor

    def set_processor(self, p):
        # diffusers pops a replaced nn.Module processor out of _modules; assigning does the same here
        if "processor" in self._modules and not isinstance(p, nn.Module):
            self._modules.pop("processor")
        self.processor = p

    def prepare_attention_mask(self, attention_mask=None, *a, **k):
        if attention_mask is None:
            return None
        raise NotImplementedError("attention masks are not used on the SDXL path")

    # the three helpers only the reference's legacy (torch < 2) processors call
    # (ip_adapter/attention_processor.py:111-119,205-224); diffusers 0.30.0 semantics with upcast_* off
    def head_to_batch_dim(self, t):
        b, l, c = t.shape
        return t.reshape(b, l, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, l, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, l, d = t.shape
        return t.reshape(bh // self.heads, self.heads, l, d).permute(0, 2, 1, 3).reshape(bh // self.heads, l, d * self.heads)

    def get_attention_scores(self, query, key, attention_mask=None):
        assert attention_mask is None
        return torch.softmax(torch.bmm(query, key.transpose(-1, -2)) * self.scale, dim=-1)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, cross_attention_dim=cross_attention_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, h, encoder_hidden_states):
        h = self.attn1(self.norm1(h)) + h
        h = self.attn2(self.norm2(h), encoder_hidden_states=encoder_hidden_states) + h
        return self.ff(self.norm3(h)) + h


class Transformer2DModel(nn.Module):
    """use_linear_projection=True variant."""

    def __init__(self, channels, heads, n_layers, cross_attention_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(channels, channels)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, cross_attention_dim) for _ in range(n_layers)])
        self.proj_out = nn.Linear(channels, channels)

    def forward(self, x, encoder_hidden_states):
        b, c, hh, ww = x.shape
        res = x
        h = self.norm(x).permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, encoder_hidden_states)
        h = self.proj_out(h)
        return h.reshape(b, hh, ww, c).permute(0, 3, 1, 2) + res


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(self.dropout(F.silu(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    """DownBlock2D (n_tf == 0) / CrossAttnDownBlock2D."""

    def __init__(self, cin, cout, n_res, n_tf, heads, cfg, add_down):
        super().__init__()
        if n_tf:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, n_tf, cfg.cross_attention_dim, cfg.norm_num_groups)
                 for _ in range(n_res)])
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps)
             for i in range(n_res)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None
        self.has_attn = bool(n_tf)

    def forward(self, h, temb, ehs):
        outs = []
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.has_attn:
                h = self.attentions[i](h, ehs)
            outs.append(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs.append(h)
        return h, outs


class MidBlock(nn.Module):
    def __init__(self, c, n_tf, heads, cfg):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, n_tf, cfg.cross_attention_dim,
                                                            cfg.norm_num_groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps)
                                      for _ in range(2)])

    def forward(self, h, temb, ehs):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, ehs)
        return self.resnets[1](h, temb)


class UpBlock(nn.Module):
    """UpBlock2D (n_tf == 0) / CrossAttnUpBlock2D."""

    def __init__(self, cin, cout, cprev, n_res, n_tf, heads, cfg, add_up):
        super().__init__()
        if n_tf:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, n_tf, cfg.cross_attention_dim, cfg.norm_num_groups)
                 for _ in range(n_res)])
        res = []
        for i in range(n_res):
            skip = cin if i == n_res - 1 else cout
            rin = cprev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None
        self.has_attn = bool(n_tf)

    def forward(self, h, skips, temb, ehs):
        for i, r in enumerate(self.resnets):
            h = torch.cat([h, skips.pop()], dim=1)
            h = r(h, temb)
            if self.has_attn:
                h = self.attentions[i](h, ehs)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg: UNetConfig = None):
        super().__init__()
        cfg = cfg or sdxl_config()
        self.config = cfg
        boc = cfg.block_out_channels
        nb = len(boc)
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], cfg.time_embed_dim)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, cfg.time_embed_dim)
        # registration order down_blocks, up_blocks, mid_block (diffusers creates both
        # ModuleLists before assigning mid_block) -> fixes the attn_processors order
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        out = boc[0]
        for i in range(nb):
            cin, out = out, boc[i]
            n_tf = 0 if i == 0 else cfg.transformer_layers_per_block[i]
            self.down_blocks.append(DownBlock(cin, out, cfg.layers_per_block, n_tf, cfg.attention_head_dim[i],
                                              cfg, add_down=(i != nb - 1)))
        self.mid_block = MidBlock(boc[-1], cfg.transformer_layers_per_block[-1], cfg.attention_head_dim[-1], cfg)
        rev = list(reversed(boc))
        rev_tf = list(reversed(cfg.transformer_layers_per_block))
        rev_heads = list(reversed(cfg.attention_head_dim))
        out = rev[0]
        for i in range(nb):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, nb - 1)]
            n_tf = 0 if i == nb - 1 else rev_tf[i]
            self.up_blocks.append(UpBlock(cin, out, prev, cfg.layers_per_block + 1, n_tf, rev_heads[i], cfg,
                                          add_up=(i != nb - 1)))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    # ---- processor protocol (diffusers API used at ip_adapter/ip_adapter.py:102,125) ----
    @property
    def attn_processors(self) -> Dict[str, object]:
        procs = {}

        def rec(name, mod):
            if hasattr(mod, "get_processor"):
                procs[f"{name}.processor"] = mod.get_processor()
            for sub, child in mod.named_children():
                rec(f"{name}.{sub}", child)

        for name, child in self.named_children():
            rec(name, child)
        return procs

    def set_attn_processor(self, processor):
        count = len(self.attn_processors)
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} "
                             f"does not match the number of attention layers: {count}.")

        def rec(name, mod):
            if hasattr(mod, "set_processor"):
                mod.set_processor(processor if not isinstance(processor, dict)
                                  else processor.pop(f"{name}.processor"))
            for sub, child in mod.named_children():
                rec(f"{name}.{sub}", child)

        if isinstance(processor, dict):
            processor = dict(processor)
        for name, child in self.named_children():
            rec(name, child)

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None,
                cross_attention_kwargs=None, return_dict=False):
        cfg = self.config
        b = sample.shape[0]
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=sample.device)
        t = t.reshape(-1).expand(b)
        t_emb = timestep_embedding(t, cfg.block_out_channels[0]).to(sample.dtype)
        emb = self.time_embedding(t_emb)
        text_embeds = added_cond_kwargs["text_embeds"]
        time_ids = added_cond_kwargs["time_ids"]
        te = timestep_embedding(time_ids.flatten(), cfg.addition_time_embed_dim).reshape(b, -1)
        add = torch.cat([text_embeds, te.to(text_embeds.dtype)], dim=-1).to(emb.dtype)
        emb = emb + self.add_embedding(add)

        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, emb, encoder_hidden_states)
            skips.extend(outs)
        h = self.mid_block(h, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            h = blk(h, skips, emb, encoder_hidden_states)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return (h,)
