"""HIP attention processors behind the diffusers AttentionProcessor protocol.

Drop-in replacements for the reference's ``AttnProcessor2_0`` / ``IPAttnProcessor2_0``
(ip_adapter/attention_processor.py:244-332, :335-465): same constructor arguments, same
attributes (``hidden_size, cross_attention_dim, scale, num_tokens, skip, to_k_ip, to_v_ip``),
same state-dict keys, same ``__call__(attn, hidden_states, encoder_hidden_states=None,
attention_mask=None, temb=None)`` contract (token-major ``[B, L, C]`` in and out), installed
through ``unet.set_attn_processor`` exactly like the reference does (ip_adapter/ip_adapter.py:99-125).

The arithmetic runs in libimh_hip.so: MFMA GEMMs for the projections, the flash kernel of
csrc/attention.hip for self-attention, and the fused to_q + text / image-prompt cross-attention
kernel of csrc/xattn.hip.  There is no torch fallback.

Two entry points per processor:
  * ``__call__``  -- the eager plugin protocol (one call = one attention layer).
  * ``emit``      -- records the same ops into a ``Ctx`` plan; used by the fused UNet forward, with the
                     text / image-prompt K,V taken from a per-image cache (they do not depend on the
                     denoise step; the reference recomputes them 30x).
"""
import os

import torch
import torch.nn as nn

from . import lib as L
from .ctx import Ctx

HEAD_DIM = 64
# the wave-specialised projection pair of self-attention (imh_gemm_dual variant 24128).  Off by default: measured 45.7 us per
# launch in the forward against 36.9 for the two-stage 128-row tiles (profiles/r03_forward_ab_dualws.json) -- its 416 workgroups
# run as two rounds of one workgroup per CU, and both forms are bound by the same L2 -> LDS rate
DUAL_WS = os.environ.get("IMH_DUAL_WS", "0") != "0"
# [Q|K|V] of a self-attention layer as ONE wave-specialised launch (imh_gemm_args.Yt: the V third leaves the kernel transposed);
# widths it is used for (A/B: IMH_QKV_ONE=0 keeps the two-problem launch; IMH_QKV_ONE_WIDTHS=1280 restricts it)
QKV_ONE = os.environ.get("IMH_QKV_ONE", "1") != "0"
QKV_ONE_WIDTHS = tuple(int(v) for v in os.environ.get("IMH_QKV_ONE_WIDTHS", "640,1280").split(",") if v)


def _pad64(n):
    return (n + 63) // 64 * 64


def _w(attn_lin, ctx):
    """weight of an nn.Linear-like holder in the compute dtype, contiguous, on the device"""
    w = attn_lin.weight
    if w.dtype != ctx.dtype or not w.is_contiguous() or w.device != ctx.device:
        w = w.detach().to(device=ctx.device, dtype=ctx.dtype).contiguous()
    return w.detach()


def _b(attn_lin, ctx):
    b = getattr(attn_lin, "bias", None)
    if b is None:
        return None
    if b.dtype != ctx.dtype or b.device != ctx.device:
        b = b.detach().to(device=ctx.device, dtype=ctx.dtype)
    return b.detach().contiguous()


def _vkey(*tensors):
    """cache key of derived (packed / folded) weights: storage address AND in-place version counter of every source tensor,
    so load_state_dict / LoRA fuse / weight.copy_ after the first forward rebuild the derived copy instead of leaving a
    stale one behind (the address alone does not change on an in-place update)"""
    return tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors)


def _packed_qk(attn, ctx):
    """[Wq; Wk] stacked so Q and K of a self-attention layer come out of ONE GEMM; cached on the module."""
    key = (_vkey(attn.to_q.weight, attn.to_k.weight), ctx.dtype, str(ctx.device))
    cached = getattr(attn, "_imh_qk", None)
    if cached is None or cached[0] != key:
        w = torch.cat([attn.to_q.weight.detach(), attn.to_k.weight.detach()], 0).to(device=ctx.device, dtype=ctx.dtype)
        cached = (key, w.contiguous())
        attn._imh_qk = cached
    return cached[1]


def fold_ln(w, norm, ctx):
    """LayerNorm folded into the following Linear:  LN(x) W^T = rstd * (x (W*gamma)^T - mean * s) + c  with
    s = sum_k (W*gamma)[., k] (taken from the ROUNDED folded weight, so the mean term cancels exactly) and
    c = W beta.  Returns (W*gamma in the compute dtype, s fp32, c fp32)."""
    g, b = norm.weight.detach().float().to(ctx.device), norm.bias.detach().float().to(ctx.device)
    wf = w.detach().float().to(ctx.device)
    wg = (wf * g[None, :]).to(ctx.dtype).contiguous()
    return wg, wg.float().sum(1).contiguous(), (wf @ b).contiguous()


def _cached(obj, name, key, make):
    c = getattr(obj, name, None)
    if c is None or c[0] != key:
        c = (key, make())
        setattr(obj, name, c)
    return c[1]


def _check_attn(attn, hidden_states, attention_mask):
    if attention_mask is not None:
        raise NotImplementedError("attention masks are not used on the SDXL path (attention_processor.py:283-287)")
    if getattr(attn, "spatial_norm", None) is not None or getattr(attn, "group_norm", None) is not None \
            or getattr(attn, "norm_cross", None):
        raise NotImplementedError("spatial_norm / group_norm / norm_cross are dead branches for SDXL")
    if hidden_states.ndim != 3:
        raise NotImplementedError("4-D hidden states are a dead branch for SDXL (attention_processor.py:379-381)")
    inner = attn.to_q.weight.shape[0]
    if inner != attn.heads * HEAD_DIM or abs(float(getattr(attn, "scale", HEAD_DIM ** -0.5)) - HEAD_DIM ** -0.5) > 1e-6:
        # the kernels index heads as h*64 and fold scale = 1/8 into the exponent: any other head width (SD-1.x:
        # 40 / 80 / 160) would silently use the wrong scale and read past the head -- refuse instead
        raise NotImplementedError(f"head_dim {inner // max(attn.heads, 1)} (scale {getattr(attn, 'scale', None)}): the HIP "
                                  f"attention kernels are built for head_dim {HEAD_DIM} (SDXL) only")


class KVCache:
    """Projected keys / values of one cross-attention layer in the layouts csrc/xattn.hip wants:
    K [B, Lk_pad, C] row-major with the 64 dims of every head stored in 16-groups ordered [0-3, 8-11, 4-7, 12-15]
    (the order the fused kernel's projected query leaves the MFMA accumulators in), Vt [C, B*Lk_pad] (transposed,
    16-key groups permuted the same way)."""
    __slots__ = ("k", "vt", "lk", "lk_pad", "k2", "vt2", "lk2", "lk2_pad")

    def __init__(self):
        self.k = self.vt = self.k2 = self.vt2 = None
        self.lk = self.lk_pad = self.lk2 = self.lk2_pad = 0


def project_kv(ctx, tokens, wk, wv):
    """tokens [B, n, Cx] -> (K [B, n_pad, C], Vt [C, B*n_pad], n, n_pad).  Padding rows are zero."""
    B, n, cx = tokens.shape
    n_pad = _pad64(n)
    x = torch.zeros(B, n_pad, cx, dtype=ctx.dtype, device=ctx.device)       # plumbing: zero-padded copy
    x[:, :n] = tokens.to(device=ctx.device, dtype=ctx.dtype)
    x2 = x.view(B * n_pad, cx)
    k = ctx.gemm(x2, wk, flags=L.GF_VT_PERM, descr="to_k")                      # [B*n_pad, C], head dims permuted
    vt = ctx.gemm(wv, x2, flags=L.GF_VT_PERM, descr="to_v^T")                   # [C, B*n_pad]
    if ctx.record:
        ctx.keep.append(x)
    return k.view(B, n_pad, -1), vt, n, n_pad


class AttnProcessor2_0(nn.Module):
    """Self-attention (reference: attention_processor.py:244-332)."""

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    # -- recorded / fused path ------------------------------------------------------------
    def emit(self, ctx, attn, x, B, L_, residual=None, kv=None, step=None, lk=None, ln=None, ln_stats=None, want_stats=False):
        """x: [B*L, C].  Returns to_out(attention(x)) (+ residual).
        ln = norm module: x is the UN-normalised residual stream and that LayerNorm is folded into the projections;
        ln_stats = (tensor, slots): the rows' statistics as left by the GEMM that wrote x (csrc/imh_lnstats.h), None -> taken
        supplied by a row-statistics launch (Ctx.gemm_dual).  Without ln, x is already layer-normed.
        want_stats: also return the row statistics of the result (the next LayerNorm's input) -> (out, stats).
        lk < L_: only the first lk rows of every batch are real keys (zero-padded sequence)."""
        C_ = x.shape[1]
        H = attn.heads
        if L_ % 64 and lk is None:
            return self._emit_ragged(ctx, attn, x, B, L_, residual, ln)
        if ln is None:
            wqk, wv = _packed_qk(attn, ctx), _w(attn.to_v, ctx)
            g1, g2 = dict(x=x, w=wqk), dict(x=wv, w=x, flags=L.GF_VT_PERM)
        else:
            norm = ln
            key = (_vkey(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, norm.weight, norm.bias), ctx.dtype, str(ctx.device))
            fq, fv = _cached(attn, "_imh_ln_qkv", key, lambda: (
                fold_ln(torch.cat([attn.to_q.weight.detach(), attn.to_k.weight.detach()], 0), norm, ctx),
                fold_ln(attn.to_v.weight, norm, ctx)))
            g1 = dict(x=x, w=fq[0], flags=L.GF_LN_ROW, ln=(fq[1], fq[2], norm.eps, ln_stats))
            g2 = dict(x=fv[0], w=x, flags=L.GF_VT_PERM | L.GF_LN_COL, ln=(fv[1], fv[2], norm.eps, ln_stats))
        if ln is not None and ln_stats is not None and QKV_ONE and (B * L_) % 256 == 0 and C_ % 160 == 0 and C_ in QKV_ONE_WIDTHS:
            # [Q|K|V] = LN(x) [Wq;Wk;Wv]^T as ONE wave-specialised launch of 256 x 160 tiles (M = 2048, N = 3840: 192 tiles, one
            # round); the V third leaves the kernel transposed through LDS, in the V^T layout the attention's PV operand reads
            f3 = _cached(attn, "_imh_ln_qkv3", key, lambda: fold_ln(
                torch.cat([attn.to_q.weight.detach(), attn.to_k.weight.detach(), attn.to_v.weight.detach()], 0), norm, ctx))
            vt = ctx.new(C_, B * L_)
            cfg3 = ctx.tuning.get((B * L_, 3 * C_, C_, 0, 1))         # (a table entry must be a wave-specialised bn = 160 variant or 23256 x 128)
            ok3 = cfg3 is not None and cfg3[2] == 1 and (B * L_) % 256 == 0 and (
                (cfg3[0] in (23256, 24128, 2464, 1464) and cfg3[1] == 160) or (tuple(cfg3[:2]) == (23256, 128) and C_ % 128 == 0))
            if not ok3:
                cfg3 = (23256, 160, 1)
            qk = ctx.gemm(x, f3[0], flags=L.GF_LN_ROW, ln=(f3[1], f3[2], norm.eps, ln_stats), cfg=tuple(cfg3), yt=(vt, 2 * C_),
                          descr="self.to_qkv")
        else:
            # [Q|K] = x [Wq;Wk]^T  [M, 2C]  and  V^T = Wv x^T  [C, M]  share x: ONE launch -- with handed-over LayerNorm statistics
            # the wave-specialised pair (variant 24128: 128 x 160 + 128 x 128 tiles), else the two-stage 128-row tiles
            ws = ln is not None and ln_stats is not None and DUAL_WS and L.experimental() and (B * L_) % 128 == 0 and C_ % 160 == 0      # the wave-specialised two-problem launch is an -DIMH_EXPERIMENTAL kernel: fall back to (128, 64) on the default library
            qk, vt = ctx.gemm_dual(g1, g2, cfg=(24128, 160) if ws else (128, 64), descr="self.to_qk+v^T")
        ao = ctx.new(B * L_, C_)
        ctx.attention(qk[:, :C_], qk[:, C_:], vt, ao, B, H, L_, lk or L_, L_, 2 * C_, 2 * C_, B * L_, C_,
                      HEAD_DIM ** -0.5, descr="self.attn")
        out = ctx.gemm(ao, _w(attn.to_out[0], ctx), bias=_b(attn.to_out[0], ctx), residual=residual,
                       descr="self.to_out", stats_out=want_stats)
        ctx.free(qk); ctx.free(vt); ctx.free(ao)
        return out

    def _emit_ragged(self, ctx, attn, x, B, L_, residual, ln):
        """Token counts that are not a multiple of the 64-key tile (e.g. 1152x896 -> 36x28 = 1008 tokens at the
        deepest level): every batch gets its own 64-padded slab of [Q|K] and V^T (dedicated zero-initialised
        buffers, so the padded keys stay finite for ever; they are masked in the kernel), projections run per batch."""
        if ln is not None:
            raise L.ImhError("ragged token counts are not supported together with the folded-LayerNorm path")
        if L_ % 16:
            raise L.ImhError(f"self-attention over {L_} tokens: the fused path needs a multiple of 16 "
                             f"(the V^T layout permutes keys in groups of 16); use a resolution whose latent sides are even")
        C_ = x.shape[1]
        H = attn.heads
        Lp = _pad64(L_)
        wqk, wv, wo, bo = _packed_qk(attn, ctx), _w(attn.to_v, ctx), _w(attn.to_out[0], ctx), _b(attn.to_out[0], ctx)
        qk = torch.zeros(B * Lp, 2 * C_, dtype=ctx.dtype, device=ctx.device)
        vt = torch.zeros(C_, B * Lp, dtype=ctx.dtype, device=ctx.device)
        ao = torch.zeros(B * Lp, C_, dtype=ctx.dtype, device=ctx.device)
        if ctx.record:
            ctx.keep.extend((qk, vt, ao))
        for b in range(B):
            xb = x[b * L_:(b + 1) * L_]
            ctx.gemm(xb, wqk, out=qk[b * Lp:b * Lp + L_], descr="self.to_qk")
            ctx.gemm(wv, xb, out=vt[:, b * Lp:b * Lp + L_], flags=L.GF_VT_PERM, descr="self.to_v^T")
        ctx.attention(qk[:, :C_], qk[:, C_:], vt, ao, B, H, Lp, L_, Lp, 2 * C_, 2 * C_, B * Lp, C_,
                      HEAD_DIM ** -0.5, descr="self.attn")
        out = ctx.new(B * L_, C_)
        for b in range(B):
            ctx.gemm(ao[b * Lp:b * Lp + L_], wo, bias=bo, out=out[b * L_:(b + 1) * L_],
                     residual=None if residual is None else residual[b * L_:(b + 1) * L_], descr="self.to_out")
        return out

    # -- eager plugin protocol ------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, **kwargs):
        _check_attn(attn, hidden_states, attention_mask)
        if encoder_hidden_states is not None:
            raise NotImplementedError("AttnProcessor2_0 with encoder_hidden_states: use IPAttnProcessor2_0(skip=True)")
        ctx = Ctx(hidden_states.device, hidden_states.dtype)
        B, L_, C_ = hidden_states.shape
        Lp = _pad64(L_)
        if Lp != L_:      # ragged sequence: zero-pad the rows, mask the padded keys (plumbing copy)
            xp = torch.zeros(B, Lp, C_, dtype=hidden_states.dtype, device=hidden_states.device)
            xp[:, :L_] = hidden_states
        else:
            xp = hidden_states.contiguous()
        x = xp.view(B * Lp, C_)
        res = x if attn.residual_connection else None
        out = self.emit(ctx, attn, x, B, Lp, residual=res, lk=L_).view(B, Lp, C_)[:, :L_]
        if attn.rescale_output_factor != 1.0:
            out = out / attn.rescale_output_factor
        return out


class IPAttnProcessor2_0(nn.Module):
    """Decoupled text + image-prompt cross-attention (reference: attention_processor.py:335-465)."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4, skip=False):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.num_tokens = num_tokens
        self.skip = skip
        self.store_attn_map = False     # True: keep the reference's `attn_map` side output (eager __call__ only)
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)

    def prepare_kv(self, ctx, attn, encoder_hidden_states):
        """Step-invariant part: K/V of the text tokens (and of the image tokens if not skip).
        The last ``num_tokens`` tokens are always sliced off, even when skip (attention_processor.py:402-406)."""
        end = encoder_hidden_states.shape[1] - self.num_tokens
        text, ip = encoder_hidden_states[:, :end], encoder_hidden_states[:, end:]
        kv = KVCache()
        kv.k, kv.vt, kv.lk, kv.lk_pad = project_kv(ctx, text, _w(attn.to_k, ctx), _w(attn.to_v, ctx))
        if not self.skip:
            kv.k2, kv.vt2, kv.lk2, kv.lk2_pad = project_kv(ctx, ip, _w(self.to_k_ip, ctx), _w(self.to_v_ip, ctx))
        return kv

    def emit(self, ctx, attn, x, B, L_, residual=None, kv=None, step=None, scale_tab=None, ln=None, ln_stats=None,
             want_stats=False):
        """ln / ln_stats / want_stats as in AttnProcessor2_0.emit"""
        C_ = x.shape[1]
        H = attn.heads
        if ln is None:
            wq, lnq = _w(attn.to_q, ctx), None
        else:       # x is the un-normalised stream; LayerNorm `ln` folded into to_q inside the fused kernel
            norm = ln
            key = (_vkey(attn.to_q.weight, norm.weight, norm.bias), ctx.dtype, str(ctx.device))
            fq = _cached(attn, "_imh_ln_q", key, lambda: fold_ln(attn.to_q.weight, norm, ctx))
            wq, lnq = fq[0], (fq[1], fq[2], norm.eps, ln_stats)
        ao = ctx.new(B * L_, C_)
        # to_q + text attention (+ image-prompt attention + text + scale * ip) in ONE launch (csrc/xattn.hip)
        if kv.k2 is not None:
            ctx.cross_attention(x, wq, kv.k, kv.vt, ao, B, H, L_, kv.lk, kv.lk_pad, C_, B * kv.lk_pad, HEAD_DIM ** -0.5, ln=lnq,
                                k2=kv.k2, vt2=kv.vt2, Lk2=kv.lk2, Lk2_pad=kv.lk2_pad, ldk2=C_, ldvt2=B * kv.lk2_pad,
                                scale2=float(self.scale), scale2_tab=scale_tab, step=step if scale_tab is not None else None,
                                descr="cross.fused+ip")
        else:
            ctx.cross_attention(x, wq, kv.k, kv.vt, ao, B, H, L_, kv.lk, kv.lk_pad, C_, B * kv.lk_pad, HEAD_DIM ** -0.5, ln=lnq,
                                descr="cross.fused")
        out = ctx.gemm(ao, _w(attn.to_out[0], ctx), bias=_b(attn.to_out[0], ctx), residual=residual,
                       descr="cross.to_out", stats_out=want_stats)
        ctx.free(ao)
        return out

    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        _check_attn(attn, hidden_states, attention_mask)
        if encoder_hidden_states is None:
            raise NotImplementedError("IPAttnProcessor2_0 is installed on cross-attention layers only "
                                      "(ip_adapter/ip_adapter.py:113-123)")
        ctx = Ctx(hidden_states.device, hidden_states.dtype)
        B, L_, C_ = hidden_states.shape
        x = hidden_states.contiguous().view(B * L_, C_)
        kv = self.prepare_kv(ctx, attn, encoder_hidden_states)
        res = x if attn.residual_connection else None
        out = self.emit(ctx, attn, x, B, L_, residual=res, kv=kv).view(B, L_, C_)
        if attn.rescale_output_factor != 1.0:
            out = out / attn.rescale_output_factor
        if getattr(self, "store_attn_map", False) and not self.skip:
            # Visualisation hook of the reference (attention_processor.py:443-444, read and deleted by
            # utils.py:9-11): q @ softmax(k_ip^T) -- upstream's operator precedence puts the softmax on k_ip^T.
            # Off by default (an extra [B, H, L, T] tensor per layer per step); plain device matmuls, not on
            # the fused / graph path.
            H = attn.heads
            ip = encoder_hidden_states[:, encoder_hidden_states.shape[1] - self.num_tokens:]
            q = torch.nn.functional.linear(hidden_states, attn.to_q.weight.to(hidden_states.dtype))
            ik = torch.nn.functional.linear(ip, self.to_k_ip.weight.to(hidden_states.dtype))
            qh = q.view(B, L_, H, C_ // H).transpose(1, 2)
            ikh = ik.view(B, -1, H, C_ // H).transpose(1, 2)
            self.attn_map = qh @ ikh.transpose(-2, -1).softmax(dim=-1)
        return out


class CNAttnProcessor2_0:
    """ControlNet processor (reference: attention_processor.py:534-621; installed by ip_adapter.py:127-133 on every
    attention layer of a ControlNet): self-attention unchanged, cross-attention on the text tokens only -- the last
    ``num_tokens`` (image-prompt) tokens are sliced off.  No parameters, like upstream."""

    def __init__(self, num_tokens=4):
        self.num_tokens = num_tokens
        self.skip = True            # prepare_kv / emit below read these three like an IPAttnProcessor2_0(skip=True)
        self.scale = 0.0

    prepare_kv = IPAttnProcessor2_0.prepare_kv
    emit = IPAttnProcessor2_0.emit

    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *args, **kwargs):
        if encoder_hidden_states is None:
            return AttnProcessor2_0()(attn, hidden_states, None, attention_mask, temb)
        return IPAttnProcessor2_0.__call__(self, attn, hidden_states, encoder_hidden_states, attention_mask, temb)


# the reference aliases these names when torch >= 2 (ip_adapter/ip_adapter.py:13-24); the isinstance()
# checks in set_scale (ip_adapter.py:181, custom_pipelines.py:19,320) go through them
AttnProcessor = AttnProcessor2_0
IPAttnProcessor = IPAttnProcessor2_0
CNAttnProcessor = CNAttnProcessor2_0
