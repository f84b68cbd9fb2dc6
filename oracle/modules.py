"""Oracle restatement of the reference's own conditioning / attention modules.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain PyTorch; every class
cites the reference lines it follows.  Parameter names are identical to the
reference's so that state dicts can be exchanged with the verbatim reference
modules (that is how tests/golden/*.pt were produced, see oracle/gen_golden.py).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# attention processors  (ip_adapter/attention_processor.py)
# --------------------------------------------------------------------------
def _heads(x, b, h):
    # [B, S, H*d] -> [B, H, S, d]     (attention_processor.py:416-419)
    return x.view(b, -1, h, x.shape[-1] // h).transpose(1, 2)


def _sdpa(q, k, v):
    # softmax(q k^T / sqrt(d)) v, fp32 math in the input dtype's precision rules of
    # F.scaled_dot_product_attention (attention_processor.py:312,423,440)
    return F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)


class AttnProcessor2_0(nn.Module):
    """Self / plain attention.  Follows attention_processor.py:244-332 for the
    SDXL case (3-D input, no spatial/group norm, no mask)."""

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, **kwargs):
        residual = hidden_states
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)                                   # :292
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states  # :294-297
        k = attn.to_k(ctx)                                             # :299
        v = attn.to_v(ctx)                                             # :300
        h = attn.heads
        o = _sdpa(_heads(q, b, h), _heads(k, b, h), _heads(v, b, h))   # :305-314
        o = o.transpose(1, 2).reshape(b, -1, q.shape[-1]).to(q.dtype)  # :316-317
        o = attn.to_out[0](o)                                          # :320
        o = attn.to_out[1](o)                                          # :322
        if attn.residual_connection:                                   # :327
            o = o + residual
        return o / attn.rescale_output_factor                          # :330


class IPAttnProcessor2_0(nn.Module):
    """Decoupled text / image-prompt cross attention.
    Follows attention_processor.py:335-465 (ctor :349-362, call :364-465)."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4, skip=False):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.num_tokens = num_tokens
        self.skip = skip
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)  # :361
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)  # :362

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual = hidden_states
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)                                   # :396
        if encoder_hidden_states is None:                              # :398
            text, ip = hidden_states, None
        else:                                                          # :402-406 (always sliced, even if skip)
            end = encoder_hidden_states.shape[1] - self.num_tokens
            text, ip = encoder_hidden_states[:, :end], encoder_hidden_states[:, end:]
        k = attn.to_k(text)                                            # :410
        v = attn.to_v(text)                                            # :411
        h = attn.heads
        qh = _heads(q, b, h)
        o = _sdpa(qh, _heads(k, b, h), _heads(v, b, h))                # :423
        o = o.transpose(1, 2).reshape(b, -1, q.shape[-1]).to(q.dtype)  # :427-428
        if not self.skip:                                              # :430
            ik = _heads(self.to_k_ip(ip), b, h)                        # :432,435
            iv = _heads(self.to_v_ip(ip), b, h)                        # :433,436
            io = _sdpa(qh, ik, iv)                                     # :440
            # side effect kept for parity of the attribute surface (:443-444); note the
            # precedence quirk: Q @ softmax_T(K_ip^T), not attention probabilities.
            with torch.no_grad():
                self.attn_map = qh @ ik.transpose(-2, -1).softmax(dim=-1)
            io = io.transpose(1, 2).reshape(b, -1, q.shape[-1]).to(q.dtype)   # :447-448
            o = o + self.scale * io                                    # :450
        o = attn.to_out[0](o)                                          # :453
        o = attn.to_out[1](o)                                          # :455
        if attn.residual_connection:                                   # :460
            o = o + residual
        return o / attn.rescale_output_factor                          # :463


# --------------------------------------------------------------------------
# Harmony-aware module  (train.py:188-266, attention_processor.py:12-56)
# --------------------------------------------------------------------------
class Cross_Attention(nn.Module):
    """attention_processor.py:12-56: head_dim = query_dim // heads (:22), score
    scale 1/sqrt(head_dim) (:23,45), softmax in the input dtype (:46)."""

    def __init__(self, query_dim, context_dim, heads=8, value_dim=None, out_dim=None):
        super().__init__()
        self.heads = heads
        self.head_dim = query_dim // heads
        self.value_dim = value_dim if value_dim is not None else self.head_dim
        self.out_dim = out_dim if out_dim is not None else heads * self.value_dim
        self.to_q = nn.Linear(query_dim, heads * self.head_dim)
        self.to_k = nn.Linear(context_dim, heads * self.head_dim)
        self.to_v = nn.Linear(context_dim, heads * self.value_dim)
        self.out_proj = nn.Linear(heads * self.value_dim, self.out_dim)

    def forward(self, query_input, context_input):
        b = query_input.size(0)
        q = self.to_q(query_input).view(b, -1, self.heads, self.head_dim).transpose(1, 2)
        k = self.to_k(context_input).view(b, -1, self.heads, self.head_dim).transpose(1, 2)
        v = self.to_v(context_input).view(b, -1, self.heads, self.value_dim).transpose(1, 2)
        p = F.softmax(q @ k.transpose(-2, -1) / math.sqrt(self.head_dim), dim=-1)
        o = (p @ v).transpose(1, 2).contiguous().view(b, -1, self.heads * self.value_dim)
        return self.out_proj(o)


class HarmonyAttention(nn.Module):
    """train.py:188-266, ``fusion_method='cross_attention'`` only (the one training and
    test.py use, train.py:582 / test.py:78; the other fusion baselines crash with the
    shipped config).  No debug prints (train.py:209,258,260)."""

    def __init__(self, image_hidden_size=1280, text_context_dim=2048, inter_dim=2560, cross_heads=10,
                 reshape_blocks=8, cross_value_dim=64, scale=1.0, fusion_method="cross_attention"):
        super().__init__()
        if fusion_method != "cross_attention":
            raise NotImplementedError("oracle restates the cross_attention fusion only")
        self.scale = scale
        self.reshape_blocks = reshape_blocks
        self.cross_query_dim = inter_dim // reshape_blocks
        self.fc1 = nn.Linear(image_hidden_size, inter_dim)                      # :208
        self.fusion_text_image = Cross_Attention(self.cross_query_dim, text_context_dim,
                                                 heads=cross_heads, value_dim=cross_value_dim)  # :210-217
        flat = cross_value_dim * cross_heads * reshape_blocks                   # :237
        self.ln = nn.LayerNorm(flat)                                            # :238
        self.fc2 = nn.Linear(flat, image_hidden_size)                           # :239

    def forward(self, text_embeds, image_embeds):
        b = image_embeds.size(0)
        x = self.fc1(image_embeds).view(b, self.reshape_blocks, self.cross_query_dim)   # :254-255
        a = self.fusion_text_image(x, text_embeds).view(b, -1)                          # :259,262
        return self.fc2(self.ln(a)) * self.scale                                        # :263-264


class ImageProjModel(nn.Module):
    """ip_adapter/ip_adapter.py:28-48."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def forward(self, image_embeds):
        t = self.proj(image_embeds).reshape(-1, self.clip_extra_context_tokens, self.cross_attention_dim)
        return self.norm(t)


class MLPProjModel(nn.Module):
    """ip_adapter/ip_adapter.py:50-66 (IPAdapterFull's projection): Linear -> GELU(erf) -> Linear -> LayerNorm, applied
    to every CLIP hidden-state token."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024):
        super().__init__()
        self.proj = nn.Sequential(nn.Linear(clip_embeddings_dim, clip_embeddings_dim), nn.GELU(),
                                  nn.Linear(clip_embeddings_dim, cross_attention_dim), nn.LayerNorm(cross_attention_dim))

    def forward(self, image_embeds):
        return self.proj(image_embeds)


# --------------------------------------------------------------------------
# Resampler  (ip_adapter/resampler.py)
# --------------------------------------------------------------------------
def FeedForward(dim, mult=4):
    # resampler.py:13-20
    inner = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(),
                         nn.Linear(inner, dim, bias=False))


class PerceiverAttention(nn.Module):
    """resampler.py:34-78."""

    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.dim_head = dim_head
        self.heads = heads
        inner = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x, latents):
        x = self.norm1(x)                                               # :57
        latents = self.norm2(latents)                                   # :58
        b, l, _ = latents.shape
        q = self.to_q(latents)                                          # :62
        k, v = self.to_kv(torch.cat((x, latents), dim=-2)).chunk(2, dim=-1)   # :63-64
        hsplit = lambda t: t.view(b, t.shape[1], self.heads, -1).transpose(1, 2)  # :23-31
        q, k, v = hsplit(q), hsplit(k), hsplit(v)
        s = 1 / math.sqrt(math.sqrt(self.dim_head))                     # :71
        w = (q * s) @ (k * s).transpose(-2, -1)                         # :72
        w = torch.softmax(w.float(), dim=-1).type(w.dtype)              # :73 (fp32 softmax)
        o = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)               # :74-76
        return self.to_out(o)                                           # :78


class Resampler(nn.Module):
    """resampler.py:81-147.  ``apply_pos_emb`` and ``num_latents_mean_pooled`` are
    kept (they are exercised by the reference's only test, test_resampler.py:18-30)."""

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768,
                 output_dim=1024, ff_mult=4, max_seq_len=257, apply_pos_emb=False,
                 num_latents_mean_pooled=0):
        super().__init__()
        self.pos_emb = nn.Embedding(max_seq_len, embedding_dim) if apply_pos_emb else None
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.num_latents_mean_pooled = num_latents_mean_pooled
        # index 1 is the Linear, as in the reference's Sequential(LayerNorm, Linear, Rearrange)
        self.to_latents_from_mean_pooled_seq = (
            nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim * num_latents_mean_pooled))
            if num_latents_mean_pooled > 0 else None)
        self.layers = nn.ModuleList([
            nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                           FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])

    def forward(self, x):
        if self.pos_emb is not None:                                    # :128-131
            x = x + self.pos_emb(torch.arange(x.shape[1], device=x.device))
        latents = self.latents.repeat(x.size(0), 1, 1)                  # :133
        x = self.proj_in(x)                                             # :135
        if self.to_latents_from_mean_pooled_seq is not None:            # :137-140
            pooled = x.mean(dim=1)                                      # masked_mean with an all-true mask
            mp = self.to_latents_from_mean_pooled_seq(pooled)
            mp = mp.view(x.size(0), self.num_latents_mean_pooled, -1)
            latents = torch.cat((mp, latents), dim=-2)
        for attn, ff in self.layers:                                    # :142-144
            latents = attn(x, latents) + latents
            latents = ff(latents) + latents
        return self.norm_out(self.proj_out(latents))                    # :146-147
