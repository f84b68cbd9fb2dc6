"""Conditioning modules of the reference on the HIP path (they run once per image, before the
denoise loop): ImageProjModel / MLPProjModel (ip_adapter/ip_adapter.py:28-66), Cross_Attention
(ip_adapter/attention_processor.py:12-56), HarmonyAttention (train.py:188-266, the
``cross_attention`` fusion that training and test.py use) and the Resampler
(ip_adapter/resampler.py:13-158).  Same constructor arguments and state-dict keys as the reference
classes; ``forward`` issues GEMM / LayerNorm / small-attention kernels through the C ABI.
torch.cat / views are data plumbing only.
"""
import math
import os

import torch
import torch.nn as nn

from . import lib as L
from .ctx import Ctx


def _ctx_for(t):
    if t.dtype not in (torch.bfloat16, torch.float16):
        raise L.ImhError(f"HIP modules compute in bf16/fp16 (got {t.dtype}); cast the inputs and the module")
    return Ctx(t.device, t.dtype)


class Graphed:
    """A conditioning module (Resampler, ImageProjModel, HarmonyAttention: ~10-45 small launches, once or twice per image) as a hipGraph:
    the first call per (input shapes, dtypes, parameter storage) runs the module once on a side stream and captures a second run; later
    calls copy the inputs into the captured buffers and replay -- 0.8 ms instead of an eager pass whose host time wanders between 0.8 and
    7 ms (BENCH_r04 `resampler`).  Outputs are clones (the captured buffers are overwritten by the next call).

    What a replay re-reads and what it freezes: parameters and buffers are read by the captured kernels at replay time through the
    pointers they had at capture (in-place updates are seen; the key below covers their storage pointers AND versions, so a re-assigned or
    in-place-modified tensor re-captures).  Plain Python attributes, and anything a module DERIVES from its weights and caches (folded /
    packed weights), are baked into the graph: the three wrapped modules derive nothing -- a module that does must not be wrapped (or must
    bump a parameter's version when the cache changes).  The capture uses the default (global) capture mode and therefore assumes that no
    other host thread allocates on this device meanwhile.  ``IMH_GRAPHED=0`` in the environment turns the wrapper into a plain eager call."""

    def __init__(self, module):
        self.module = module
        self._cache = {}

    def __getattr__(self, name):                 # state_dict / parameters / attributes of the wrapped module
        return getattr(self.__dict__["module"], name)

    @torch.no_grad()
    def __call__(self, *xs):
        if not xs or any((not torch.is_tensor(x)) or x.device.type != "cuda" for x in xs) or os.environ.get("IMH_GRAPHED", "1") == "0":
            return self.module(*xs)
        key = tuple((tuple(x.shape), x.dtype, x.device.index) for x in xs) + \
            tuple((t.data_ptr(), t._version) for t in list(self.module.parameters()) + list(self.module.buffers()))
        ent = self._cache.get(key)
        if ent is None:
            dev = xs[0].device
            # (the adapters call this under torch.inference_mode(): the capture machinery updates generator state in place and the
            # captured buffers are written by every later call, so both are made outside inference mode, as normal tensors)
            with torch.inference_mode(False), torch.no_grad():
                ins = [x.detach().clone() for x in xs]
                cur, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
                side.wait_stream(cur)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    self.module(*ins)            # warm-up: one-time attribute calls / allocations stay out of the capture
                    torch.cuda.synchronize(dev)
                    with torch.cuda.graph(g, stream=side):
                        out = self.module(*ins)
                cur.wait_stream(side)
            self._cache.clear()                  # one live graph per module (parameters reloaded -> new key)
            ent = self._cache[key] = (ins, g, out)
        for d, x in zip(ent[0], xs):
            d.copy_(x)
        ent[1].replay()
        return ent[2].clone()


def _lin(ctx, lin, x, **kw):
    return ctx.gemm(x, lin.weight.detach(), bias=None if lin.bias is None else lin.bias.detach(), **kw)


def _ln(ctx, norm, x):
    return ctx.layernorm(x, norm.weight.detach(), norm.bias.detach(), norm.eps)


class ImageProjModel(nn.Module):
    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4):
        super().__init__()
        self.generator = None
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    @torch.no_grad()
    def forward(self, image_embeds):
        ctx = _ctx_for(image_embeds)
        x = image_embeds.reshape(-1, image_embeds.shape[-1]).contiguous()
        t = _lin(ctx, self.proj, x).view(-1, self.cross_attention_dim)
        return _ln(ctx, self.norm, t).view(-1, self.clip_extra_context_tokens, self.cross_attention_dim)


class MLPProjModel(nn.Module):
    """IPAdapterFull's projection (ip_adapter/ip_adapter.py:50-66): Linear -> GELU(erf) -> Linear -> LayerNorm on every
    CLIP hidden-state token; the GELU rides in the first GEMM's epilogue."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024):
        super().__init__()
        self.proj = nn.Sequential(nn.Linear(clip_embeddings_dim, clip_embeddings_dim), nn.GELU(),
                                  nn.Linear(clip_embeddings_dim, cross_attention_dim), nn.LayerNorm(cross_attention_dim))

    @torch.no_grad()
    def forward(self, image_embeds):
        ctx = _ctx_for(image_embeds)
        x = image_embeds.reshape(-1, image_embeds.shape[-1]).contiguous()
        h = _lin(ctx, self.proj[0], x, flags=L.GF_ACT_GELU)
        t = _lin(ctx, self.proj[2], h)
        return _ln(ctx, self.proj[3], t).view(*image_embeds.shape[:-1], -1)


class Cross_Attention(nn.Module):
    """head_dim = query_dim // heads (attention_processor.py:22), scores / sqrt(head_dim) (:23,45)."""

    def __init__(self, query_dim, context_dim, heads=8, value_dim=None, out_dim=None):
        super().__init__()
        self.query_dim = query_dim
        self.heads = heads
        self.head_dim = query_dim // heads
        self.scale = math.sqrt(self.head_dim)
        self.value_dim = value_dim if value_dim is not None else self.head_dim
        self.out_dim = out_dim if out_dim is not None else heads * self.value_dim
        self.to_q = nn.Linear(query_dim, heads * self.head_dim)
        self.to_k = nn.Linear(context_dim, heads * self.head_dim)
        self.to_v = nn.Linear(context_dim, heads * self.value_dim)
        self.out_proj = nn.Linear(heads * self.value_dim, self.out_dim)

    @torch.no_grad()
    def forward(self, query_input, context_input):
        ctx = _ctx_for(query_input)
        B = query_input.size(0)
        qi = query_input.reshape(B, -1, query_input.shape[-1])
        ci = context_input.reshape(B, -1, context_input.shape[-1])     # view(B, -1, ...) semantics (:40-42)
        Lq, Lk = qi.shape[1], ci.shape[1]
        q = _lin(ctx, self.to_q, qi.reshape(B * Lq, -1).contiguous())
        k = _lin(ctx, self.to_k, ci.reshape(B * Lk, -1).contiguous())
        v = _lin(ctx, self.to_v, ci.reshape(B * Lk, -1).contiguous())
        o = ctx.attention_small(q, k, v, B, self.heads, Lq, Lk, self.head_dim, self.value_dim, 1.0 / self.scale)
        return _lin(ctx, self.out_proj, o).view(B, Lq, self.out_dim)


class HarmonyAttention(nn.Module):
    """train.py:188-266 with fusion_method='cross_attention' (the ablation fusions of baseline.py crash with
    the shipped config and are out of scope).  No debug prints."""

    def __init__(self, image_hidden_size=1280, text_context_dim=2048, inter_dim=2560, cross_heads=10,
                 reshape_blocks=8, cross_value_dim=64, scale=1.0, fusion_method="cross_attention"):
        super().__init__()
        if fusion_method != "cross_attention":
            raise NotImplementedError("only the 'cross_attention' fusion is on the IMAGHarmony hot path")
        self.scale = scale
        self.reshape_blocks = reshape_blocks
        self.cross_query_dim = inter_dim // reshape_blocks
        self.fusion_method = fusion_method
        self.image_hidden_size = image_hidden_size
        self.text_context_dim = text_context_dim
        self.fc1 = nn.Linear(image_hidden_size, inter_dim)
        self.fusion_text_image = Cross_Attention(self.cross_query_dim, text_context_dim, heads=cross_heads,
                                                 value_dim=cross_value_dim)
        flat = cross_value_dim * cross_heads * reshape_blocks
        self.ln = nn.LayerNorm(flat)
        self.fc2 = nn.Linear(flat, image_hidden_size)

    @torch.no_grad()
    def forward(self, text_embeds, image_embeds):
        ctx = _ctx_for(image_embeds)
        B = image_embeds.size(0)
        x = _lin(ctx, self.fc1, image_embeds.contiguous()).view(B, self.reshape_blocks, self.cross_query_dim)
        a = self.fusion_text_image(x, text_embeds.to(image_embeds.dtype)).reshape(B, -1)
        out = _lin(ctx, self.fc2, _ln(ctx, self.ln, a.contiguous()))
        return out if self.scale == 1.0 else out * self.scale


def FeedForward(dim, mult=4):
    inner = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(), nn.Linear(inner, dim, bias=False))


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.dim_head = dim_head
        self.heads = heads
        inner = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def emit(self, ctx, x, latents, B, n1, n2):
        """x [B*n1, D], latents [B*n2, D] -> to_out(attn) + latents  (resampler.py:49-78 and the residual :143)"""
        inner = self.dim_head * self.heads
        xn, ln_ = _ln(ctx, self.norm1, x), _ln(ctx, self.norm2, latents)
        q = _lin(ctx, self.to_q, ln_)
        kv_in = torch.cat([xn.view(B, n1, -1), ln_.view(B, n2, -1)], dim=1).reshape(B * (n1 + n2), -1)   # :63
        kv = _lin(ctx, self.to_kv, kv_in)
        # (q*s)(k*s)^T with s = d^-1/4 (:71-72) == q k^T * d^-1/2; softmax in fp32 (:73)
        o = ctx.attention_small(q, kv[:, :inner], kv[:, inner:], B, self.heads, n2, n1 + n2, self.dim_head,
                                self.dim_head, self.dim_head ** -0.5)
        return _lin(ctx, self.to_out, o, residual=latents)


class Resampler(nn.Module):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, max_seq_len: int = 257, apply_pos_emb: bool = False, num_latents_mean_pooled: int = 0):
        super().__init__()
        self.pos_emb = nn.Embedding(max_seq_len, embedding_dim) if apply_pos_emb else None
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.num_latents_mean_pooled = num_latents_mean_pooled
        self.to_latents_from_mean_pooled_seq = (
            nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim * num_latents_mean_pooled))
            if num_latents_mean_pooled > 0 else None)
        self.layers = nn.ModuleList([
            nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads), FeedForward(dim=dim, mult=ff_mult)])
            for _ in range(depth)])

    @torch.no_grad()
    def forward(self, x):
        ctx = _ctx_for(x)
        B, n1, _ = x.shape
        if self.pos_emb is not None:
            x = x + self.pos_emb.weight[:n1].to(x.dtype)                 # plumbing-level add of a table (:128-131)
        dim = self.latents.shape[-1]
        lat = self.latents.detach().to(x.dtype).repeat(B, 1, 1)
        xp = _lin(ctx, self.proj_in, x.reshape(B * n1, -1).contiguous())
        if self.to_latents_from_mean_pooled_seq is not None:
            # masked_mean with an all-true mask == mean over the sequence (:137-140): a [1/n1] row-vector GEMM per batch
            pooled = xp.view(B, n1, dim).float().mean(1).to(x.dtype)
            mp = _lin(ctx, self.to_latents_from_mean_pooled_seq[1],
                      _ln(ctx, self.to_latents_from_mean_pooled_seq[0], pooled.contiguous()))
            lat = torch.cat([mp.view(B, self.num_latents_mean_pooled, dim), lat], dim=1)
        n2 = lat.shape[1]
        lat = lat.reshape(B * n2, dim).contiguous()
        for attn, ff in self.layers:
            lat = attn.emit(ctx, xp, lat, B, n1, n2)
            h = _lin(ctx, ff[1], _ln(ctx, ff[0], lat), flags=L.GF_ACT_GELU)
            lat = _lin(ctx, ff[3], h, residual=lat)
        out = _ln(ctx, self.norm_out, _lin(ctx, self.proj_out, lat))
        return out.view(B, n2, -1)
