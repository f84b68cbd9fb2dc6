"""SDXL ``UNet2DConditionModel`` forward on the gfx950 kernels.

What the reference calls at ip_adapter/custom_pipelines.py:338-345 / train.py:310 is diffusers' UNet
(third-party; architecture per SURVEY.md Appendix A).  This module keeps its contract -- the SDXL
state-dict key schema, ``.config``, ``.attn_processors``, ``.set_attn_processor(dict)``, and
``forward(sample, timestep, encoder_hidden_states, added_cond_kwargs=..., return_dict=False)[0]`` --
but the arithmetic is emitted as calls into libimh_hip.so:

  * activations live token-major / NHWC ([B, H*W, C]) for the whole network, so the
    NCHW<->token permutes of Transformer2DModel disappear and every 3x3 conv is an implicit GEMM
    with a contiguous K axis; NCHW exists only at conv_in's input and after conv_out;
  * GEGLU, bias, residual adds, time-embedding adds and nearest-upsampling are epilogue / gather
    options of the GEMM kernel, not separate passes;
  * the 17 ``time_emb_proj`` Linears are stacked into ONE GEMM per step;
  * text / image-prompt K,V of the 70 cross-attention layers come from a per-image cache.

torch is used for parameter storage and device memory only.  ``emit_forward`` records the whole
forward into a ``Ctx`` (plan / hipGraph); ``forward`` is the eager drop-in signature.
"""
import math
import os
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import lib as L
from .attention_processor import AttnProcessor2_0, IPAttnProcessor2_0, _b, _vkey, _w
from .ctx import Ctx, GnSpec


# BasicTransformerBlock.norm1/2/3 folded into the consumer GEMMs (exact algebra, tested): the consumer takes the
# un-normalised residual stream and normalises in its epilogue, y = rstd * (acc - mean * s) + c -- 210 LayerNorm launches and
# 0.84 GB of HBM traffic per forward disappear.  False = the stand-alone LayerNorm kernel in front of every consumer (A/B and
# debugging).
FOLD_LAYERNORM = True
# The rows' (mean, rstd) travel with the residual stream: the GEMM that WRITES a LayerNorm input (proj_in, to_out + residual,
# ff.out + residual) leaves (sum, M2) slot partials behind from its epilogue (csrc/imh_lnstats.h) and every consumer -- the
# [Q|K] + V^T pair (norm1), the fused cross-attention's to_q (norm2), ff.net.0 (norm3) -- merges them in its prologue:
# Welford / Chan all the way, like torch.nn.LayerNorm (which the reference runs ahead of attn.to_q,
# ip_adapter/attention_processor.py:396).  The in-loop E[x^2] - mean^2 form of rounds 2-3 no longer exists in the library;
# False here (IMH_LN_STATS=0, A/B) makes every consumer take its statistics from a stand-alone row-statistics launch instead.
LN_STATS_HANDOVER = os.environ.get("IMH_LN_STATS", "1") != "0"
# GroupNorm statistics the same way: the conv / GEMM that writes a GroupNorm input (conv1 -> norm2, conv2 / proj_out / the
# stride-2 downsampler -> the next block's norm1 / Transformer2DModel.norm / conv_norm_out, ALSO through the up path's channel
# concats: the two producers' partials are merged) leaves (sum, M2) partials per 10-channel sub-run behind from its epilogue
# (imh_gemm_args.gn_out); a tiny launch turns them into the per-sample (scale, shift) table of the GroupNorm.  Tensors no epilogue
# covers (conv_in, split-K / ring variants) get theirs from one statistics pass.  False = every tensor takes the pass (A/B;
# IMH_GN_STATS=0).
GN_STATS_HANDOVER = os.environ.get("IMH_GN_STATS", "1") != "0"
# ... and the GroupNorm itself -- diffusers ResnetBlock2D: norm -> SiLU -> conv -- is applied INSIDE the consuming
# conv3x3's halo staging wherever that conv runs on the LDS-halo kernel (the 128 x 128 and 64 x 64 levels): no normalised tensor in
# memory, and the up path's torch.cat([hidden, skip], 1) is read from its two producers by the same kernel (and by conv_shortcut's
# GEMM): no concat pass either.  False = table + apply pass + materialised concat everywhere (A/B; IMH_GN_FUSE=0).
GN_FUSE = os.environ.get("IMH_GN_FUSE", "1") != "0"
# ... and (round 5) the table step has no launch of its own either: the consumer -- the fused conv, or the apply pass in front of a
# Linear -- builds its sample's (scale, shift) table in its prologue from the producers' partials (imh_gemm_args.gn_part,
# IMH_GN_TABLE_APPLY; csrc/imh_gntable.h: the same routine as the table launch, bit-identical).  False = one table launch per
# GroupNorm (A/B; IMH_GN_TABLE_FOLD=0).
GN_TABLE_FOLD = os.environ.get("IMH_GN_TABLE_FOLD", "1") != "0"


class Feat:
    """an activation travelling through the UNet with the GroupNorm partials its writing launch left behind (or None)"""
    __slots__ = ("t", "gs")

    def __init__(self, t, gs=None):
        self.t, self.gs = t, gs

    def stats(self, ctx, groups):
        """the tensor's GroupNorm partials: the producer's, else one statistics pass (kept: skip tensors are normalised twice).
        Sub-runs of gcd(C / groups, 10) channels tile the groups of every reader, alone or through a concat with a tensor of a
        multiple of C channels (SDXL: 10, like the producers' epilogues)"""
        if self.gs is None:
            B, C_ = self.t.shape[0], self.t.shape[-1]
            self.gs = ctx.gn_stats(self.t.view(B, -1, C_), sub=math.gcd(C_ // groups, 10))
        return self.gs

    def free(self, ctx):
        ctx.free(self.t)
        if self.gs is not None:
            ctx.free(self.gs.t)
            self.gs = None


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    sample_size: int = 128
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20)
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


# ------------------------------------------------------------------ parameter holders
class Linear(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout), requires_grad=False) if bias else None


class Conv2d(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout), requires_grad=False)

    def packed(self, ctx):
        """[Cout, Cin, k, k] -> [Cout, k*k*Cin] (K index = (ky*k+kx)*Cin + c), cached."""
        key = (_vkey(self.weight), ctx.dtype, str(ctx.device))
        c = getattr(self, "_imh_packed", None)
        if c is None or c[0] != key:
            w = self.weight.detach().permute(0, 2, 3, 1).reshape(self.weight.shape[0], -1)
            c = (key, w.to(device=ctx.device, dtype=ctx.dtype).contiguous())
            self._imh_packed = c
        return c[1]


class Norm(nn.Module):
    def __init__(self, c, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(c), requires_grad=False)
        self.eps = eps


class Dropout(nn.Module):
    pass


class Attention(nn.Module):
    """Parameter holder with the attribute surface the processors touch."""

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
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(kv, inner, bias=False)
        self.to_v = Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([Linear(inner, query_dim), Dropout()])
        self.processor = AttnProcessor2_0()

    def get_processor(self):
        return self.processor

    def set_processor(self, p):
        if "processor" in self._modules and not isinstance(p, nn.Module):
            self._modules.pop("processor")
        self.processor = p

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask)


def geglu_interleave(t):
    """[2 * inner, ...] = [values; gates] (diffusers GEGLU.proj: hidden, gate = proj(x).chunk(2)) -> rows in quads (value_2k,
    value_2k+1, gate_2k, gate_2k+1): one output tile holds both halves and the two gates of a quad land in one even-aligned
    accumulator register pair of the MFMA tile (IMH_GF_GEGLU: packed-fp32 GELU on both at once)"""
    inner = t.shape[0] // 2
    v, g = t[:inner].reshape((inner // 2, 2) + tuple(t.shape[1:])), t[inner:].reshape((inner // 2, 2) + tuple(t.shape[1:]))
    return torch.cat([v, g], 1).reshape((2 * inner,) + tuple(t.shape[1:])).contiguous()


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = Linear(dim, inner * 2)

    def packed(self, ctx):
        """rows interleaved in (value, value, gate, gate) quads (geglu_interleave; GF_GEGLU epilogue)."""
        key = (_vkey(self.proj.weight, self.proj.bias), ctx.dtype, str(ctx.device))
        c = getattr(self, "_imh_packed", None)
        if c is None or c[0] != key:
            w, b = self.proj.weight.detach(), self.proj.bias.detach()
            wi, bi = geglu_interleave(w), geglu_interleave(b)
            c = (key, wi.to(device=ctx.device, dtype=ctx.dtype).contiguous(),
                 bi.to(device=ctx.device, dtype=ctx.dtype).contiguous())
            self._imh_packed = c
        return c[1], c[2]


def _geglu_packed_ln(self, ctx, norm):
    """GEGLU projection with LayerNorm folded in (see attention_processor.fold_ln), rows interleaved in quads (geglu_interleave)."""
    from .attention_processor import fold_ln
    key = (_vkey(self.proj.weight, self.proj.bias, norm.weight, norm.bias), ctx.dtype, str(ctx.device))
    c = getattr(self, "_imh_packed_ln", None)
    if c is None or c[0] != key:
        wg, s, cc = fold_ln(self.proj.weight, norm, ctx)
        il = geglu_interleave
        # the Linear's bias rides in the fold's constant term, c = W beta + b (fp32): one epilogue operand (and twenty live
        # registers per lane of the 256 x 160 kernel's epilogue) less than a separate bias vector
        b = self.proj.bias.detach().to(device=ctx.device, dtype=torch.float32)
        c = (key, il(wg), None, il(s), il(cc + b))
        self._imh_packed_ln = c
    return c[1], c[2], c[3], c[4]


GEGLU.packed_ln = _geglu_packed_ln


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), Dropout(), Linear(dim * 4, dim)])

    def emit(self, ctx, x, residual, ln=None, ln_stats=None, want_stats=False):
        if ln is None:
            w1, b1 = self.net[0].packed(ctx)
            g = ctx.gemm(x, w1, bias=b1, flags=L.GF_GEGLU, descr="ff.geglu")
        else:       # x un-normalised, LayerNorm `ln` folded into the (interleaved) GEGLU projection
            w1, b1, s1, c1 = self.net[0].packed_ln(ctx, ln)
            g = ctx.gemm(x, w1, bias=b1, flags=L.GF_GEGLU | L.GF_LN_ROW, ln=(s1, c1, ln.eps, ln_stats), descr="ff.geglu")
        out = ctx.gemm(g, _w(self.net[2], ctx), bias=_b(self.net[2], ctx), residual=residual, descr="ff.out", stats_out=want_stats)
        ctx.free(g)
        return out


def _ln(ctx, norm, x, descr):
    return ctx.layernorm(x, _w(norm, ctx), _b(norm, ctx), norm.eps, descr=descr)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_attention_dim):
        super().__init__()
        self.norm1 = Norm(dim, 1e-5)
        self.attn1 = Attention(dim, heads)
        self.norm2 = Norm(dim, 1e-5)
        self.attn2 = Attention(dim, heads, cross_attention_dim=cross_attention_dim)
        self.norm3 = Norm(dim, 1e-5)
        self.ff = FeedForward(dim)

    def fused(self, L_):
        """the folded-LayerNorm form applies (both processors are the HIP ones, 64-aligned token count)"""
        return FOLD_LAYERNORM and isinstance(self.attn1.processor, AttnProcessor2_0) \
            and isinstance(self.attn2.processor, IPAttnProcessor2_0) and L_ % 64 == 0

    def emit(self, ctx, h, B, L_, kv, st, stats=None, want_stats=False):
        """h: the residual stream [B*L, C] (consumed).  stats = row statistics of h from the GEMM that wrote it (or None);
        want_stats: return (h3, statistics of h3) for the next block's norm1."""
        p1, p2 = self.attn1.processor, self.attn2.processor
        if not hasattr(p1, "emit") or not hasattr(p2, "emit"):
            raise L.ImhError("a non-HIP attention processor is installed; the fused forward needs "
                             "imagharmony_amd.attention_processor processors")
        ip2 = isinstance(p2, IPAttnProcessor2_0)
        if self.fused(L_):
            # LayerNorm never materialises: every consumer GEMM reads the residual stream itself, and (LN_STATS_HANDOVER)
            # takes the rows' statistics from the epilogue of the GEMM that wrote it
            ho = LN_STATS_HANDOVER
            if not ho:
                stats = None
            r = p1.emit(ctx, self.attn1, h, B, L_, residual=h, ln=self.norm1, ln_stats=stats, want_stats=ho)
            h1, s1 = r if ho else (r, None)
            ctx.free(h)
            if stats is not None:
                ctx.free(stats[0])
            r = p2.emit(ctx, self.attn2, h1, B, L_, residual=h1, kv=kv, step=st.step, scale_tab=st.ip_scale_tab, ln=self.norm2,
                        ln_stats=s1, want_stats=ho)
            h2, s2 = r if ho else (r, None)
            ctx.free(h1)
            if s1 is not None:
                ctx.free(s1[0])
            r = self.ff.emit(ctx, h2, residual=h2, ln=self.norm3, ln_stats=s2, want_stats=ho and want_stats)
            h3, s3 = r if (ho and want_stats) else (r, None)
            ctx.free(h2)
            if s2 is not None:
                ctx.free(s2[0])
            return (h3, s3) if want_stats else h3
        if stats is not None:
            ctx.free(stats[0])
        n = _ln(ctx, self.norm1, h, "norm1")
        h1 = p1.emit(ctx, self.attn1, n, B, L_, residual=h)
        ctx.free(n); ctx.free(h)
        n = _ln(ctx, self.norm2, h1, "norm2")
        h2 = p2.emit(ctx, self.attn2, n, B, L_, residual=h1, kv=kv, step=st.step, scale_tab=st.ip_scale_tab) \
            if ip2 else p2.emit(ctx, self.attn2, n, B, L_, residual=h1, kv=kv)
        ctx.free(n); ctx.free(h1)
        n = _ln(ctx, self.norm3, h2, "norm3")
        h3 = self.ff.emit(ctx, n, residual=h2)
        ctx.free(n); ctx.free(h2)
        return (h3, None) if want_stats else h3


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, n_layers, cross_attention_dim, groups):
        super().__init__()
        self.norm = Norm(channels, 1e-6)
        self.groups = groups
        self.proj_in = Linear(channels, channels)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, cross_attention_dim) for _ in range(n_layers)])
        self.proj_out = Linear(channels, channels)

    def emit(self, ctx, f, kvs, st):
        """f: Feat of NHWC [B, H, W, C] (consumed) -> Feat of the same shape, with the GroupNorm partials of the output left by
        proj_out's epilogue (None if its variant has none)."""
        x = f.t
        B, Hh, Ww, C_ = x.shape
        L_ = Hh * Ww
        x2 = x.view(B * L_, C_)
        if GN_TABLE_FOLD:
            n = ctx.gn_table_apply(x.view(B, L_, C_), GnSpec(f.stats(ctx, self.groups), _w(self.norm, ctx), _b(self.norm, ctx), self.groups, self.norm.eps),
                                   False, descr="t2d.norm")
        else:
            tab = ctx.gn_table(f.stats(ctx, self.groups), _w(self.norm, ctx), _b(self.norm, ctx), self.groups, self.norm.eps, L_, descr="t2d.norm.table")
            n = ctx.gn_apply(x.view(B, L_, C_), tab, False, descr="t2d.norm")
            ctx.free(tab)
        blocks = list(self.transformer_blocks)
        ho = LN_STATS_HANDOVER and bool(blocks) and blocks[0].fused(L_)
        r = ctx.gemm(n.view(B * L_, C_), _w(self.proj_in, ctx), bias=_b(self.proj_in, ctx), descr="t2d.proj_in", stats_out=ho)
        h, stats = r if ho else (r, None)
        ctx.free(n)
        for i, (blk, kv) in enumerate(zip(blocks, kvs)):
            more = ho and i + 1 < len(blocks) and blocks[i + 1].fused(L_)
            r = blk.emit(ctx, h, B, L_, kv, st, stats=stats, want_stats=more)
            h, stats = r if more else (r, None)
        r = ctx.gemm(h, _w(self.proj_out, ctx), bias=_b(self.proj_out, ctx), residual=x2, descr="t2d.proj_out",
                     gn_out=L_ if GN_STATS_HANDOVER else None)
        out, gn = r if GN_STATS_HANDOVER else (r, None)
        ctx.free(h)
        f.free(ctx)
        return Feat(out.view(B, Hh, Ww, C_), gn)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.groups = groups
        self.norm1 = Norm(cin, eps)
        self.conv1 = Conv2d(cin, cout, 3)
        self.time_emb_proj = Linear(temb_dim, cout)
        self.norm2 = Norm(cout, eps)
        self.conv2 = Conv2d(cout, cout, 3)
        self.conv_shortcut = Conv2d(cin, cout, 1) if cin != cout else None
        self.temb_offset = 0    # column of this block inside the stacked time_emb_proj output

    def emit(self, ctx, f, st, skip=None, keep_input=False):
        """f: Feat of NHWC [B, H, W, C1]; skip: Feat of the skip connection (the block's input is torch.cat([f, skip], channels),
        diffusers UpBlock2D / CrossAttnUpBlock2D) or None -> Feat [B, H, W, Cout].  Both inputs are consumed unless keep_input.
        GroupNorm statistics travel with the tensors (Feat.gs); where the convs run on the LDS-halo kernel, norm1 / norm2 (+ SiLU)
        and the concat happen inside them (GN_FUSE), else as passes."""
        x = f.t
        B, Hh, Ww, C1 = x.shape
        HW = Hh * Ww
        M = B * HW
        sk = skip.t if skip is not None else None
        Cin = C1 + (sk.shape[-1] if sk is not None else 0)
        Cout = self.conv1.weight.shape[0]
        want = 1 if GN_STATS_HANDOVER else 0
        srcs = [f.stats(ctx, self.groups)] + ([skip.stats(ctx, self.groups)] if skip is not None else [])
        ra = st.temb_all[:, self.temb_offset:self.temb_offset + Cout]
        fuse1 = GN_FUSE and ctx.conv_fuses_gn(M, Cout, 9 * Cin)
        spec1 = GnSpec(srcs, _w(self.norm1, ctx), _b(self.norm1, ctx), self.groups, self.norm1.eps)
        # the table: built by the consumer itself (fold), or by one small launch
        tab1 = spec1 if GN_TABLE_FOLD else ctx.gn_table(srcs, _w(self.norm1, ctx), _b(self.norm1, ctx), self.groups, self.norm1.eps, HW, descr="res.norm1.table")
        xc = None                       # the materialised concat, where something still needs it
        if fuse1:
            h = ctx.conv3x3(x, self.conv1.packed(ctx), bias=_b(self.conv1, ctx), rowadd=ra, ldra=st.temb_all.stride(0),
                            descr="res.conv1", gn_groups=want, gn=(tab1, True), x2=sk)
        else:
            xc = ctx.concat(x, sk, descr="skip.concat") if sk is not None else x
            n = (ctx.gn_table_apply(xc.view(B, HW, Cin), tab1, True, descr="res.norm1") if GN_TABLE_FOLD
                 else ctx.gn_apply(xc.view(B, HW, Cin), tab1, True, descr="res.norm1")).view(B, Hh, Ww, Cin)
            h = ctx.conv3x3(n, self.conv1.packed(ctx), bias=_b(self.conv1, ctx), rowadd=ra, ldra=st.temb_all.stride(0),
                            descr="res.conv1", gn_groups=want)
            ctx.free(n)
        if not GN_TABLE_FOLD:
            ctx.free(tab1)
        hf = Feat(*h) if want else Feat(h)
        tab2 = (GnSpec(hf.stats(ctx, self.groups), _w(self.norm2, ctx), _b(self.norm2, ctx), self.groups, self.norm2.eps) if GN_TABLE_FOLD
                else ctx.gn_table(hf.stats(ctx, self.groups), _w(self.norm2, ctx), _b(self.norm2, ctx), self.groups, self.norm2.eps, HW, descr="res.norm2.table"))
        if self.conv_shortcut is not None:
            wsc, bsc = self.conv_shortcut.packed(ctx), _b(self.conv_shortcut, ctx)
            if sk is not None and xc is None:
                sc = ctx.gemm(x.view(M, C1), wsc, bias=bsc, x2=sk.view(M, Cin - C1), descr="res.shortcut")
            else:
                sc = ctx.gemm((xc if xc is not None else x).view(M, Cin), wsc, bias=bsc, descr="res.shortcut")
        else:
            # identity shortcut: the residual is the block's INPUT -- with a skip connection that is the concatenated tensor (no SDXL up
            # block gets here: Cin = C1 + Cskip != Cout always has a conv_shortcut)
            if sk is not None and xc is None:
                xc = ctx.concat(x, sk, descr="skip.concat")
            sc = (xc if xc is not None else x).view(M, Cin)
        if GN_FUSE and ctx.conv_fuses_gn(M, Cout, 9 * Cout):
            out = ctx.conv3x3(hf.t, self.conv2.packed(ctx), bias=_b(self.conv2, ctx), residual=sc, descr="res.conv2", gn_groups=want,
                              gn=(tab2, True))
        else:
            n = (ctx.gn_table_apply(hf.t.view(B, HW, Cout), tab2, True, descr="res.norm2") if GN_TABLE_FOLD
                 else ctx.gn_apply(hf.t.view(B, HW, Cout), tab2, True, descr="res.norm2")).view(B, Hh, Ww, Cout)
            out = ctx.conv3x3(n, self.conv2.packed(ctx), bias=_b(self.conv2, ctx), residual=sc, descr="res.conv2", gn_groups=want)
            ctx.free(n)
        if not GN_TABLE_FOLD:
            ctx.free(tab2)
        hf.free(ctx)
        if self.conv_shortcut is not None:
            ctx.free(sc)
        if xc is not None and sk is not None:
            ctx.free(xc)
        if not keep_input:
            f.free(ctx)
            if skip is not None:
                skip.free(ctx)
        return Feat(*out) if want else Feat(out)


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv2d(c, c, 3)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv2d(c, c, 3)


class DownBlock(nn.Module):
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


class MidBlock(nn.Module):
    def __init__(self, c, n_tf, heads, cfg):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, n_tf, cfg.cross_attention_dim,
                                                            cfg.norm_num_groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps)
                                      for _ in range(2)])


class UpBlock(nn.Module):
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


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = Linear(in_dim, dim)
        self.linear_2 = Linear(dim, dim)


class StepState:
    """Device-resident per-image / per-step state the recorded forward reads."""

    def __init__(self):
        self.latents = None        # fp32 [S, 4, H, W] (NCHW) -- scheduler state
        self.t_table = None        # fp32 [steps] timesteps
        self.step = None           # int32 [1] device step counter
        self.in_scale_tab = None   # fp32 [steps] scale_model_input factor (Euler) or None
        self.ip_scale_tab = None   # fp32 [steps] IP scale per step (control_guidance gating) or None
        self.aug_emb = None        # [B, time_embed_dim]  add_embedding(text_embeds ++ time_ids) (step-invariant)
        self.kv = None             # {attn2 processor name: KVCache}
        self.temb_all = None       # [B, sum Cout] stacked time_emb_proj output (per step)
        self.temb_table = None     # [steps, B * sum Cout] the same for EVERY step of the schedule (precompute_temb), or None
        self.t_value = None        # fp32 [B] explicit timestep values (eager forward)


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg: UNetConfig = None):
        super().__init__()
        cfg = cfg or UNetConfig()
        self.config = cfg
        boc = cfg.block_out_channels
        nb = len(boc)
        self.conv_in = Conv2d(cfg.in_channels, boc[0], 3)
        self.time_embedding = TimestepEmbedding(boc[0], cfg.time_embed_dim)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, cfg.time_embed_dim)
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])     # created before mid_block: fixes the attn_processors order
        out = boc[0]
        for i in range(nb):
            cin, out = out, boc[i]
            n_tf = 0 if i == 0 else cfg.transformer_layers_per_block[i]
            self.down_blocks.append(DownBlock(cin, out, cfg.layers_per_block, n_tf, cfg.attention_head_dim[i], cfg,
                                              add_down=(i != nb - 1)))
        self.mid_block = MidBlock(boc[-1], cfg.transformer_layers_per_block[-1], cfg.attention_head_dim[-1], cfg)
        rev, rev_tf = list(reversed(boc)), list(reversed(cfg.transformer_layers_per_block))
        rev_heads = list(reversed(cfg.attention_head_dim))
        out = rev[0]
        for i in range(nb):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, nb - 1)]
            n_tf = 0 if i == nb - 1 else rev_tf[i]
            self.up_blocks.append(UpBlock(cin, out, prev, cfg.layers_per_block + 1, n_tf, rev_heads[i], cfg,
                                          add_up=(i != nb - 1)))
        self.conv_norm_out = Norm(boc[0], cfg.norm_eps)
        self.conv_out = Conv2d(boc[0], cfg.out_channels, 3)
        # stacked time_emb_proj bookkeeping
        off = 0
        for r in self.resnets():
            r.temb_offset = off
            off += r.time_emb_proj.weight.shape[0]
        self.temb_total = off

    # ---- iteration helpers ----
    def resnets(self):
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            for r in blk.resnets:
                yield r

    # ---- processor protocol (ip_adapter/ip_adapter.py:102,125) ----
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
        if isinstance(processor, dict):
            processor = dict(processor)

        def rec(name, mod):
            if hasattr(mod, "set_processor"):
                mod.set_processor(processor if not isinstance(processor, dict) else processor.pop(f"{name}.processor"))
            for sub, child in mod.named_children():
                rec(f"{name}.{sub}", child)

        for name, child in self.named_children():
            rec(name, child)

    def attn2_modules(self):
        """[(processor name, Attention)] for the cross-attention layers, in attn_processors order."""
        out = []

        def rec(name, mod):
            if isinstance(mod, Attention) and name.endswith("attn2"):
                out.append((f"{name}.processor", mod))
            for sub, child in mod.named_children():
                rec(f"{name}.{sub}", child)

        for name, child in self.named_children():
            rec(name, child)
        return out

    # ---- random init directly on the device (synthetic-weight benchmarks) ----
    @torch.no_grad()
    def init_random_(self, seed=1234):
        g = torch.Generator(device=self.conv_in.weight.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if p.ndim >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32).mul_(fan_in ** -0.5))
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32).mul_(0.02))
            else:
                p.fill_(1.0)
        return self

    # ---- checkpoints (the on-disk format either side of the path) ----
    @classmethod
    def from_safetensors(cls, path, cfg: "UNetConfig" = None, device="cpu", dtype=None):
        """Load a diffusers-format SDXL UNet checkpoint (``diffusion_pytorch_model.safetensors``): the key schema is
        the one this module keeps, so this is a strict ``load_state_dict``."""
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
        m = cls(cfg)
        own = m.state_dict()
        extra = [k for k in sd if k not in own]
        missing = [k for k in own if k not in sd and ".processor." not in k]
        if extra or missing:
            raise KeyError(f"checkpoint / model key mismatch: {len(missing)} missing (e.g. {missing[:3]}), "
                           f"{len(extra)} unexpected (e.g. {extra[:3]})")
        m.load_state_dict(sd, strict=False)
        m = m.to(device)
        return m.to(dtype) if dtype is not None else m

    # ---- packed / stacked weights ----
    def _temb_stack(self, ctx):
        key = (_vkey(*[t for r in self.resnets() for t in (r.time_emb_proj.weight, r.time_emb_proj.bias)]), ctx.dtype, str(ctx.device))
        c = getattr(self, "_imh_temb", None)
        if c is None or c[0] != key:
            w = torch.cat([r.time_emb_proj.weight.detach() for r in self.resnets()], 0)
            b = torch.cat([r.time_emb_proj.bias.detach() for r in self.resnets()], 0)
            c = (key, w.to(device=ctx.device, dtype=ctx.dtype).contiguous(), b.to(device=ctx.device, dtype=ctx.dtype))
            self._imh_temb = c
        return c[1], c[2]

    def _temb_chain(self, ctx, tsin, aug):
        """time_embedding(t) + aug_emb -> SiLU -> the 17 stacked time_emb_proj Linears: [rows, 320] -> [rows, sum Cout]"""
        te = self.time_embedding
        h = ctx.gemm(tsin, _w(te.linear_1, ctx), bias=_b(te.linear_1, ctx), flags=L.GF_ACT_SILU, descr="time_emb.1")
        # emb = time_embedding(t) + aug_emb, then SiLU (every ResnetBlock2D applies it before time_emb_proj)
        emb = ctx.gemm(h, _w(te.linear_2, ctx), bias=_b(te.linear_2, ctx), residual=aug, descr="time_emb.2")
        semb = ctx.silu(emb, descr="silu(emb)")
        wt, bt = self._temb_stack(ctx)
        out = ctx.gemm(semb, wt, bias=bt, descr="time_emb_proj(all)")
        ctx.free(h); ctx.free(emb); ctx.free(semb)
        return out

    @torch.no_grad()
    def precompute_temb(self, ctx, st, timesteps):
        """st.temb_table = the stacked time-embedding projections of EVERY step of a schedule ([steps, B * sum Cout]; row i = what
        the per-step chain yields at timestep i): they depend on the step and on the conditioning (aug_emb), not on the latent,
        so a denoise loop computes them once per (schedule, conditioning) instead of five launches per step (r03 DESIGN 9.4)."""
        B = st.aug_emb.shape[0]
        ts = timesteps.to(device=ctx.device, dtype=torch.float32).reshape(-1)
        n = ts.numel()
        tv = ts.repeat_interleave(B).contiguous()                      # row i * B + b -> timestep i
        tsin = ctx.new(n * B, self.config.block_out_channels[0])
        ctx.ew(L.EW_TIMESTEP, tsin, a=tv, n=n * B, i=(self.config.block_out_channels[0], 0, 0, 0, 0, 0), descr="time_proj(all steps)")
        aug = st.aug_emb.repeat(n, 1).contiguous()                     # plumbing: the residual rows of every step
        tab = self._temb_chain(ctx, tsin, aug)
        ctx.free(tsin)
        st.temb_table = tab.view(n, B * self.temb_total)
        if ctx.record:
            ctx.keep.extend([tv, aug])
        return st.temb_table

    # ---- step-invariant conditioning ----
    @torch.no_grad()
    def prepare_conditioning(self, ctx, encoder_hidden_states, text_embeds, time_ids, st=None):
        """aug_emb = add_embedding(cat[text_embeds, Timesteps(256)(time_ids)]) and the K/V cache of every
        cross-attention layer (all independent of the denoise step)."""
        st = st or StepState()
        cfg = self.config
        B = encoder_hidden_states.shape[0]
        ids = time_ids.to(device=ctx.device, dtype=torch.float32).reshape(-1).contiguous()
        te = ctx.new(ids.numel(), cfg.addition_time_embed_dim)
        ctx.ew(L.EW_TIMESTEP, te, a=ids, n=ids.numel(), i=(cfg.addition_time_embed_dim, 0, 0, 0, 0, 0),
               descr="add_time_proj")
        add_in = torch.cat([text_embeds.to(device=ctx.device, dtype=ctx.dtype), te.view(B, -1)], dim=-1).contiguous()
        ae = self.add_embedding
        h = ctx.gemm(add_in, _w(ae.linear_1, ctx), bias=_b(ae.linear_1, ctx), flags=L.GF_ACT_SILU, descr="add_emb.1")
        st.aug_emb = ctx.gemm(h, _w(ae.linear_2, ctx), bias=_b(ae.linear_2, ctx), descr="add_emb.2")
        ehs = encoder_hidden_states.to(device=ctx.device, dtype=ctx.dtype)
        st.kv = {}
        for name, attn in self.attn2_modules():
            proc = attn.processor
            if not hasattr(proc, "prepare_kv"):
                raise L.ImhError(f"{name}: processor {type(proc).__name__} has no HIP K/V path")
            st.kv[name] = proc.prepare_kv(ctx, attn, ehs)
        if ctx.record:
            ctx.keep.extend([ids, add_in, ehs])
        return st

    # ---- the forward, as emitted ops ----
    def emit_forward(self, ctx, st, S, Hl, Wl, cfg_dup=True):
        """Records one UNet forward.  Reads st.latents (fp32 NCHW [S,4,Hl,Wl]); batch B = 2S when
        cfg_dup (CFG halves share the latent, custom_pipelines.py:332).  Returns the noise prediction
        as NHWC [B, Hl*Wl, 4] in the compute dtype."""
        cfg = self.config
        B = 2 * S if cfg_dup else S
        boc = cfg.block_out_channels
        div = 1 << (len(boc) - 1)
        if Hl % div or Wl % div:
            # diffusers interpolates the up path to the skip's size in that case (forward_upsample_size); the
            # fused nearest-x2 upsampling here does not, so refuse instead of producing misaligned skips
            raise L.ImhError(f"latent {Hl}x{Wl}: sides must be multiples of {div} (image sides multiples of {8 * div})")
        # -- time embedding --
        ctx.tag = 1
        if st.temb_table is not None and st.t_table is not None and st.temb_table.shape[1] == B * self.temb_total:
            # the chain below depends on the step and the conditioning only, not on the latent: all steps' rows were computed once
            # per schedule (precompute_temb); the step's row is copied in (one launch instead of five)
            st.temb_all = ctx.new(B, self.temb_total)
            ctx.ew(L.EW_STEP_ROW, st.temb_all, a=st.temb_table, step=st.step, n=B * self.temb_total, descr="temb_row(step)",
                   nbytes=2.0 * B * self.temb_total * st.temb_all.element_size())
        else:
            tsin = ctx.new(B, boc[0])
            if st.t_table is not None:
                ctx.ew(L.EW_TIMESTEP, tsin, a=st.t_table, step=st.step, n=B, i=(boc[0], 0, 0, 0, 0, 0), descr="time_proj")
            else:
                ctx.ew(L.EW_TIMESTEP, tsin, a=st.t_value, n=B, i=(boc[0], 0, 0, 0, 0, 0), descr="time_proj")
            st.temb_all = self._temb_chain(ctx, tsin, st.aug_emb)
            ctx.free(tsin)
        # -- conv_in (+ CFG duplication + scale_model_input) --
        ctx.tag = 2
        x = ctx.new(B, Hl, Wl, boc[0])
        ctx.ew(L.EW_CONV_IN, x, a=st.latents, w=_w(self.conv_in, ctx), bias=_b(self.conv_in, ctx),
               tab=st.in_scale_tab, step=st.step if st.in_scale_tab is not None else None,
               i=(S, Hl, Wl, boc[0], B, 0), f=(1.0, 0, 0, 0), descr="conv_in",
               nbytes=2.0 * B * Hl * Wl * boc[0])
        h = Feat(x)                      # (conv_in has no statistics epilogue: Feat.stats() takes one pass, shared by both readers)
        skips = [h]
        ho = GN_STATS_HANDOVER
        kv_of = lambda t2d: [st.kv[self._pname(t2d, k)] for k in range(len(t2d.transformer_blocks))]
        # -- down --
        for bi, blk in enumerate(self.down_blocks):
            for i, r in enumerate(blk.resnets):
                ctx.tag = 10 + bi
                h = r.emit(ctx, h, st, keep_input=True)       # inputs are skip tensors: keep
                if blk.has_attn:
                    ctx.tag = 20 + bi
                    h = blk.attentions[i].emit(ctx, h, kv_of(blk.attentions[i]), st)
                skips.append(h)
            if blk.downsamplers is not None:
                ctx.tag = 10 + bi
                d = blk.downsamplers[0].conv
                r_ = ctx.conv3x3(h.t, d.packed(ctx), bias=_b(d, ctx), stride=2, descr="downsample", gn_groups=1 if ho else 0)
                h = Feat(*r_) if ho else Feat(r_)
                skips.append(h)
        # -- mid --
        ctx.tag = 30
        mb = self.mid_block
        h = mb.resnets[0].emit(ctx, h, st, keep_input=True)
        ctx.tag = 31
        h = mb.attentions[0].emit(ctx, h, kv_of(mb.attentions[0]), st)
        ctx.tag = 30
        h = mb.resnets[1].emit(ctx, h, st)
        # -- up --
        for bi, blk in enumerate(self.up_blocks):
            for i, r in enumerate(blk.resnets):
                ctx.tag = 40 + bi
                h = r.emit(ctx, h, st, skip=skips.pop())      # torch.cat([hidden, skip], 1) is read from its two producers
                if blk.has_attn:
                    ctx.tag = 50 + bi
                    h = blk.attentions[i].emit(ctx, h, kv_of(blk.attentions[i]), st)
            if blk.upsamplers is not None:
                ctx.tag = 40 + bi
                u = blk.upsamplers[0].conv
                r_ = ctx.conv3x3(h.t, u.packed(ctx), bias=_b(u, ctx), up=1, descr="upsample", gn_groups=1 if ho else 0)
                h.free(ctx)
                h = Feat(*r_) if ho else Feat(r_)
        # -- out --
        ctx.tag = 60
        Bh, Hh, Ww, C0 = h.t.shape
        if GN_TABLE_FOLD:
            n = ctx.gn_table_apply(h.t.view(B, Hh * Ww, C0), GnSpec(h.stats(ctx, cfg.norm_num_groups), _w(self.conv_norm_out, ctx), _b(self.conv_norm_out, ctx),
                                                                    cfg.norm_num_groups, cfg.norm_eps), True, descr="conv_norm_out").view(B, Hh, Ww, C0)
        else:
            tab = ctx.gn_table(h.stats(ctx, cfg.norm_num_groups), _w(self.conv_norm_out, ctx), _b(self.conv_norm_out, ctx), cfg.norm_num_groups, cfg.norm_eps,
                               Hh * Ww, descr="conv_norm_out.table")
            n = ctx.gn_apply(h.t.view(B, Hh * Ww, C0), tab, True, descr="conv_norm_out").view(B, Hh, Ww, C0)
            ctx.free(tab)
        h.free(ctx)
        out = ctx.conv3x3(n, self.conv_out.packed(ctx), bias=_b(self.conv_out, ctx), descr="conv_out")
        ctx.free(n)
        ctx.tag = 0
        return out.view(B, Hh * Ww, cfg.out_channels)


    def _pname(self, t2d, k):
        names = getattr(self, "_imh_pnames", None)
        if names is None or id(t2d.transformer_blocks[k].attn2) not in names:      # (a deepcopy carries stale ids)
            names = {}
            for name, mod in self.named_modules():
                if isinstance(mod, Attention) and name.endswith("attn2"):
                    names[id(mod)] = f"{name}.processor"
            self._imh_pnames = names
        return names[id(t2d.transformer_blocks[k].attn2)]

    # ---- eager drop-in signature (diffusers UNet2DConditionModel.forward) ----
    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, cross_attention_kwargs=None,
                return_dict=False, **kw):
        dev = sample.device
        dtype = self.conv_in.weight.dtype if self.conv_in.weight.dtype in (torch.bfloat16, torch.float16) \
            else (sample.dtype if sample.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16)
        ctx = Ctx(dev, dtype)
        B, _, Hl, Wl = sample.shape
        st = self.prepare_conditioning(ctx, encoder_hidden_states, added_cond_kwargs["text_embeds"],
                                       added_cond_kwargs["time_ids"])
        t = timestep if torch.is_tensor(timestep) else torch.tensor([float(timestep)])
        st.t_value = t.to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        st.latents = sample.to(torch.float32).contiguous()
        out = self.emit_forward(ctx, st, B, Hl, Wl, cfg_dup=False)
        y = out.view(B, Hl, Wl, -1).permute(0, 3, 1, 2).to(sample.dtype)
        return (y,)
