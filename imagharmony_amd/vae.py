"""SDXL VAE decode on the HIP path -- the row right after the denoise loop:
``image = vae.decode(latents / scaling_factor)`` (ip_adapter/custom_pipelines.py:365-379, run in fp32 upstream
because the SDXL VAE overflows in fp16; test.py:73 turns tiling on) and ``image_processor.postprocess`` (:386).

Same parameter names as diffusers' ``AutoencoderKL`` (``decoder.*``, ``post_quant_conv.*``; encoder keys are accepted
and ignored), so a real SDXL VAE state dict drops in.  Two compute modes, chosen like the reference chooses
(``needs_upcasting = vae.dtype == float16 and vae.config.force_upcast`` -> ``upcast_vae()``, custom_pipelines.py:366-372):

* **fp32** (round 6; a float16 module with ``force_upcast``, or a float32 module): fp32 activations, fp32 weights (the stored
  weights upcast exactly, as ``vae.to(float32)`` does) and fp32 arithmetic on the fp32 kernels of csrc/f32.hip
  (``v_mfma_f32_32x32x2_f32`` implicit-GEMM conv3x3 / GEMM, GroupNorm statistics merged in double, row softmax) -- the
  reference's precision, no fp16 overflow; ~0.1 s per 1024^2 image (the fp32 matrix rate is 1/16 of bf16's).
* **native** (a bfloat16 module -- which the reference does not upcast either -- or ``precision="native"``): NHWC activations
  in the module's 16-bit dtype on the UNet's own kernels -- GroupNorm(+SiLU), implicit-GEMM conv3x3 (fused nearest-x2
  upsampling), GEMM -- 18 ms per image, 1e-2 rel-RMS from the fp32 result in bf16.

Both run a materialised single-head attention for the mid block (scores GEMM -> fp32 row softmax -> PV GEMM: one head of
width 512 does not fit the head_dim-64 flash kernel, and it runs once per image).  Tiled decoding follows diffusers'
``tiled_decode`` (overlapping 64x64-latent tiles, linear blends).  Host-side torch is plumbing only: channel padding
of the 4-channel latent, the tile blends / concatenation, and the final NHWC->image conversion.
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn

from . import lib as L
from .attention_processor import _b, _vkey, _w
from .ctx import Ctx
from .unet import Conv2d, Linear, Norm


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    sample_size: int = 1024
    scaling_factor: float = 0.13025
    force_upcast: bool = True
    tile_overlap_factor: float = 0.25


def _gn(ctx, norm, x, groups, silu, descr):
    B, H, W, C_ = x.shape
    return ctx.groupnorm(x.view(B, H * W, C_), _w(norm, ctx), _b(norm, ctx), groups, norm.eps, silu=silu,
                         descr=descr).view(B, H, W, C_)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.groups = groups
        self.norm1 = Norm(cin, 1e-6)
        self.conv1 = Conv2d(cin, cout, 3)
        self.norm2 = Norm(cout, 1e-6)
        self.conv2 = Conv2d(cout, cout, 3)
        self.conv_shortcut = Conv2d(cin, cout, 1) if cin != cout else None

    def emit(self, ctx, x):
        """x NHWC [B, H, W, Cin] (consumed) -> [B, H, W, Cout]"""
        B, H, W, Cin = x.shape
        n = _gn(ctx, self.norm1, x, self.groups, True, "vae.res.norm1")
        h = ctx.conv3x3(n, self.conv1.packed(ctx), bias=_b(self.conv1, ctx), descr="vae.res.conv1")
        ctx.free(n)
        n = _gn(ctx, self.norm2, h, self.groups, True, "vae.res.norm2")
        ctx.free(h)
        if self.conv_shortcut is not None:
            sc = ctx.gemm(x.view(B * H * W, Cin), self.conv_shortcut.packed(ctx), bias=_b(self.conv_shortcut, ctx),
                          descr="vae.res.shortcut")
        else:
            sc = x.view(B * H * W, Cin)
        out = ctx.conv3x3(n, self.conv2.packed(ctx), bias=_b(self.conv2, ctx), residual=sc, descr="vae.res.conv2")
        ctx.free(n)
        if self.conv_shortcut is not None:
            ctx.free(sc)
        ctx.free(x)
        return out


class VAEAttention(nn.Module):
    """mid-block attention: heads = 1, GroupNorm over the tokens, projections with bias, residual"""

    def __init__(self, channels, groups):
        super().__init__()
        self.groups = groups
        self.group_norm = Norm(channels, 1e-6)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), nn.Identity()])

    def emit(self, ctx, x):
        B, H, W, C_ = x.shape
        Lq = H * W
        if Lq % 64:
            raise L.ImhError(f"VAE attention: {H}x{W} tokens must be a multiple of 64")
        n = _gn(ctx, self.group_norm, x, self.groups, False, "vae.attn.norm").view(B * Lq, C_)
        q = ctx.gemm(n, _w(self.to_q, ctx), bias=_b(self.to_q, ctx), descr="vae.attn.to_q")
        k = ctx.gemm(n, _w(self.to_k, ctx), bias=_b(self.to_k, ctx), descr="vae.attn.to_k")
        # V^T = Wv n^T (swapped operands) feeds the PV GEMM as its [N, K] operand; softmax rows sum to 1, so the
        # to_v bias is added once after PV instead of to every value row
        vt = ctx.gemm(_w(self.to_v, ctx), n, descr="vae.attn.to_v^T")                       # [C, B*L]
        ctx.free(n)
        o = ctx.new(B * Lq, C_)
        sc = ctx.new(Lq, Lq, dtype=torch.float32)
        pr = ctx.new(Lq, Lq)
        for b in range(B):
            qb, kb = q[b * Lq:(b + 1) * Lq], k[b * Lq:(b + 1) * Lq]
            ctx.gemm(qb, kb, out=sc, flags=L.GF_OUT_F32, descr="vae.attn.scores")           # [L, L] fp32
            ctx.ew(L.EW_SOFTMAX, pr, a=sc, i=(Lq, Lq, Lq, Lq, 0, 0), f=(C_ ** -0.5, 0.0, 0.0, 0.0),
                   descr="vae.attn.softmax", nbytes=6.0 * Lq * Lq)
            ctx.gemm(pr, vt[:, b * Lq:(b + 1) * Lq], out=o[b * Lq:(b + 1) * Lq], bias=_b(self.to_v, ctx), N=C_, K=Lq,
                     ldw=B * Lq, descr="vae.attn.pv")
        ctx.free(sc); ctx.free(pr); ctx.free(q); ctx.free(k); ctx.free(vt)
        out = ctx.gemm(o, _w(self.to_out[0], ctx), bias=_b(self.to_out[0], ctx), residual=x.view(B * Lq, C_),
                       descr="vae.attn.to_out")
        ctx.free(o); ctx.free(x)
        return out.view(B, H, W, C_)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = Conv2d(ch, ch, 3)

    def emit(self, ctx, x):
        out = ctx.conv3x3(x, self.conv.packed(ctx), bias=_b(self.conv, ctx), up=1, descr="vae.upsample")   # nearest x2 fused
        ctx.free(x)
        return out


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(n)])
        if up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def emit(self, ctx, x):
        for r in self.resnets:
            x = r.emit(ctx, x)
        for u in getattr(self, "upsamplers", []):
            x = u.emit(ctx, x)
        return x


class MidBlock(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.attentions = nn.ModuleList([VAEAttention(ch, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, groups), ResnetBlock2D(ch, ch, groups)])

    def emit(self, ctx, x):
        return self.resnets[1].emit(ctx, self.attentions[0].emit(ctx, self.resnets[0].emit(ctx, x)))


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch = tuple(reversed(cfg.block_out_channels))
        g = cfg.norm_num_groups
        self.groups = g
        self.conv_in = Conv2d(cfg.latent_channels, ch[0], 3)
        self.mid_block = MidBlock(ch[0], g)
        self.up_blocks = nn.ModuleList()
        c = ch[0]
        for i, co in enumerate(ch):
            self.up_blocks.append(UpDecoderBlock2D(c, co, cfg.layers_per_block + 1, g, up=i < len(ch) - 1))
            c = co
        self.conv_norm_out = Norm(ch[-1], 1e-6)
        self.conv_out = Conv2d(ch[-1], cfg.out_channels, 3)


def _pad_conv_in(conv, ctx, cpad=64):
    """conv_in reads a 4-channel latent: zero-pad Cin to 64 so it runs on the implicit-GEMM kernel (K = 9*64)"""
    key = (_vkey(conv.weight), ctx.dtype, str(ctx.device))
    c = getattr(conv, "_imh_padded", None)
    if c is None or c[0] != key:
        w = conv.weight.detach()
        wp = torch.zeros(w.shape[0], cpad, 3, 3, dtype=w.dtype, device=w.device)
        wp[:, :w.shape[1]] = w
        c = (key, wp.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(device=ctx.device, dtype=ctx.dtype).contiguous())
        conv._imh_padded = c
    return c[1]


def _pad_1x1(conv, ctx, cpad=64):
    key = (_vkey(conv.weight, conv.bias), ctx.dtype, str(ctx.device))
    c = getattr(conv, "_imh_padded", None)
    if c is None or c[0] != key:
        w = conv.weight.detach().view(conv.weight.shape[0], -1)
        wp = torch.zeros(cpad, cpad, dtype=w.dtype, device=w.device)
        wp[:w.shape[0], :w.shape[1]] = w
        bp = torch.zeros(cpad, dtype=w.dtype, device=w.device)
        bp[:w.shape[0]] = conv.bias.detach()
        c = (key, wp.to(device=ctx.device, dtype=ctx.dtype).contiguous(), bp.to(device=ctx.device, dtype=ctx.dtype))
        conv._imh_padded = c
    return c[1], c[2]


# ---- fp32 (reference-precision) decode: the same graph on csrc/f32.hip --------------------------------------------------
def _f32_cached(mod, name, build, *src):
    """fp32 copy of a module's (packed / padded) parameter on the device, rebuilt when the source changes (in place or re-assigned)"""
    key = (_vkey(*src), str(src[0].device))
    c = getattr(mod, name, None)
    if c is None or c[0] != key:
        c = (key, build())
        setattr(mod, name, c)
    return c[1]


def _f32_conv_w(conv, cin_pad=0):
    """[Cout, Cin, 3, 3] -> packed fp32 [Cout, 9 * Cin'] (K index = (ky*3+kx)*Cin' + c), Cin zero-padded to cin_pad"""
    def build():
        w = conv.weight.detach().float()
        if cin_pad and cin_pad > w.shape[1]:
            wp = torch.zeros(w.shape[0], cin_pad, 3, 3, dtype=torch.float32, device=w.device)
            wp[:, :w.shape[1]] = w
            w = wp
        return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
    return _f32_cached(conv, "_imh_f32_w", build, conv.weight)


def _f32_lin_w(lin):
    return _f32_cached(lin, "_imh_f32_w", lambda: lin.weight.detach().float().reshape(lin.weight.shape[0], -1).contiguous(), lin.weight)


def _f32_vec(mod, attr="bias"):
    t = getattr(mod, attr)
    return _f32_cached(mod, "_imh_f32_" + attr, lambda: t.detach().float().contiguous(), t)


def _gn32(ctx, norm, x, groups, silu, descr):
    B, H, W, C_ = x.shape
    return ctx.f32_groupnorm(x.view(B, H * W, C_), _f32_vec(norm, "weight"), _f32_vec(norm, "bias"), groups, norm.eps, silu=silu,
                             descr=descr).view(B, H, W, C_)


def _res32(ctx, r, x):
    B, H, W, Cin = x.shape
    n = _gn32(ctx, r.norm1, x, r.groups, True, "vae32.res.norm1")
    h = ctx.f32_conv3x3(n, _f32_conv_w(r.conv1), bias=_f32_vec(r.conv1), descr="vae32.res.conv1")
    ctx.free(n)
    n = _gn32(ctx, r.norm2, h, r.groups, True, "vae32.res.norm2")
    ctx.free(h)
    if r.conv_shortcut is not None:
        sc = ctx.f32_gemm(x.view(B * H * W, Cin), _f32_lin_w(r.conv_shortcut), bias=_f32_vec(r.conv_shortcut), descr="vae32.res.shortcut")
    else:
        sc = x.view(B * H * W, Cin)
    out = ctx.f32_conv3x3(n, _f32_conv_w(r.conv2), bias=_f32_vec(r.conv2), residual=sc, descr="vae32.res.conv2")
    ctx.free(n)
    if r.conv_shortcut is not None:
        ctx.free(sc)
    ctx.free(x)
    return out


def _attn32(ctx, at, x):
    B, H, W, C_ = x.shape
    Lq = H * W
    n = _gn32(ctx, at.group_norm, x, at.groups, False, "vae32.attn.norm").view(B * Lq, C_)
    q = ctx.f32_gemm(n, _f32_lin_w(at.to_q), bias=_f32_vec(at.to_q), descr="vae32.attn.to_q")
    k = ctx.f32_gemm(n, _f32_lin_w(at.to_k), bias=_f32_vec(at.to_k), descr="vae32.attn.to_k")
    # V^T = Wv n^T (swapped operands) feeds the PV GEMM as its [N, K] operand; softmax rows sum to 1, so the to_v bias is added once after PV
    vt = ctx.f32_gemm(_f32_lin_w(at.to_v), n, descr="vae32.attn.to_v^T")                      # [C, B*L]
    ctx.free(n)
    o = ctx.new(B * Lq, C_, dtype=torch.float32)
    sc = ctx.new(Lq, Lq, dtype=torch.float32)
    for b in range(B):
        qb, kb = q[b * Lq:(b + 1) * Lq], k[b * Lq:(b + 1) * Lq]
        ctx.f32_gemm(qb, kb, out=sc, descr="vae32.attn.scores")
        ctx.f32_softmax(sc, sc, C_ ** -0.5, descr="vae32.attn.softmax")                        # in place (a row is read before it is written)
        ctx.f32_gemm(sc, vt[:, b * Lq:(b + 1) * Lq], out=o[b * Lq:(b + 1) * Lq], bias=_f32_vec(at.to_v), N=C_, K=Lq, ldw=B * Lq,
                     descr="vae32.attn.pv")
    ctx.free(sc); ctx.free(q); ctx.free(k); ctx.free(vt)
    out = ctx.f32_gemm(o, _f32_lin_w(at.to_out[0]), bias=_f32_vec(at.to_out[0]), residual=x.view(B * Lq, C_), descr="vae32.attn.to_out")
    ctx.free(o); ctx.free(x)
    return out.view(B, H, W, C_)


class AutoencoderKL(nn.Module):
    """decode-only AutoencoderKL.  ``decode(z)`` returns the image tensor [B, 3, 8h, 8w] (fp32, roughly [-1, 1])."""

    def __init__(self, config: VAEConfig = None):
        super().__init__()
        self.config = config or VAEConfig()
        c = self.config
        self.decoder = Decoder(c)
        self.post_quant_conv = Conv2d(c.latent_channels, c.latent_channels, 1)
        self.use_tiling = False
        self.tile_sample_min_size = c.sample_size if c.sample_size < 512 else 512
        self.tile_latent_min_size = int(self.tile_sample_min_size / (2 ** (len(c.block_out_channels) - 1)))
        self.tile_overlap_factor = c.tile_overlap_factor

    # -- loading --
    def load_state_dict(self, sd, strict=True, **kw):
        sd = {k: v for k, v in sd.items() if not (k.startswith("encoder.") or k.startswith("quant_conv."))}
        return super().load_state_dict(sd, strict=strict, **kw)

    @classmethod
    def from_safetensors(cls, path, config: VAEConfig = None, device="cuda:0", dtype=torch.bfloat16):
        from safetensors.torch import load_file
        m = cls(config)
        m.load_state_dict(load_file(path), strict=True)
        return m.to(device, dtype)

    def init_random_(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        for n, p in self.named_parameters():
            if p.dim() > 1:
                fan = p[0].numel()
                p.data.copy_((torch.randn(p.shape, generator=g) * fan ** -0.5).to(p.dtype))
            else:
                p.data.fill_(1.0 if n.endswith("weight") else 0.0)
        return self

    def enable_tiling(self, on=True):                 # pipe.enable_vae_tiling(), test.py:73
        self.use_tiling = on

    @property
    def dtype(self):
        return self.decoder.conv_in.weight.dtype

    # -- compute --
    def precision_for(self, precision=None):
        """'fp32' | 'native': the reference's rule (custom_pipelines.py:366-372) unless the caller names one"""
        precision = precision or getattr(self, "precision", "auto")
        if precision == "auto":
            dt = self.dtype
            precision = "fp32" if dt == torch.float32 or (dt == torch.float16 and self.config.force_upcast) else "native"
        if precision not in ("fp32", "native"):
            raise ValueError(f"precision {precision!r} (expected 'auto', 'fp32' or 'native')")
        return precision

    def _decode_tile_f32(self, z):
        """z: [B, 4, h, w] fp32 on the device -> [B, 3, 8h, 8w] fp32, every activation and every product in fp32"""
        dev = z.device
        ctx = Ctx(dev, torch.bfloat16)                                           # (pool / stream only: every tensor below is fp32)
        B, _, h, w = z.shape
        d = self.decoder
        zp = torch.zeros(B, h, w, 16, dtype=torch.float32, device=dev)           # plumbing: NHWC, channels padded to the K step of 16
        zp[..., :z.shape[1]] = z.permute(0, 2, 3, 1)
        pq = self.post_quant_conv

        def build_pq():
            wq = torch.zeros(16, 16, dtype=torch.float32, device=dev)
            wq[:pq.weight.shape[0], :pq.weight.shape[1]] = pq.weight.detach().float().view(pq.weight.shape[0], -1)
            bq = torch.zeros(16, dtype=torch.float32, device=dev)
            bq[:pq.bias.shape[0]] = pq.bias.detach().float()
            return wq, bq
        wq, bq = _f32_cached(pq, "_imh_f32_w", build_pq, pq.weight, pq.bias)
        t = ctx.f32_gemm(zp.view(B * h * w, 16), wq, bias=bq, descr="vae32.post_quant").view(B, h, w, 16)
        x = ctx.f32_conv3x3(t, _f32_conv_w(d.conv_in, cin_pad=16), bias=_f32_vec(d.conv_in), descr="vae32.conv_in")
        ctx.free(t)
        x = _res32(ctx, d.mid_block.resnets[0], x)
        x = _attn32(ctx, d.mid_block.attentions[0], x)
        x = _res32(ctx, d.mid_block.resnets[1], x)
        for blk in d.up_blocks:
            for r in blk.resnets:
                x = _res32(ctx, r, x)
            for u in getattr(blk, "upsamplers", []):
                y = ctx.f32_conv3x3(x, _f32_conv_w(u.conv), bias=_f32_vec(u.conv), up=1, descr="vae32.upsample")    # nearest x2 fused
                ctx.free(x)
                x = y
        n = _gn32(ctx, d.conv_norm_out, x, d.groups, True, "vae32.conv_norm_out")
        ctx.free(x)
        y = ctx.f32_conv3x3(n, _f32_conv_w(d.conv_out), bias=_f32_vec(d.conv_out), descr="vae32.conv_out")          # [B, 8h, 8w, 3]
        ctx.free(n)
        return y.permute(0, 3, 1, 2).contiguous()

    def _decode_tile(self, z, precision="native"):
        """z: [B, 4, h, w] fp32 on the device -> [B, 3, 8h, 8w] fp32"""
        if precision == "fp32":
            return self._decode_tile_f32(z)
        dev, dt = z.device, self.dtype
        if dt not in (torch.bfloat16, torch.float16):
            raise L.ImhError("the native HIP VAE path computes in bf16 or fp16 (precision='fp32' runs any module in fp32)")
        ctx = Ctx(dev, dt)
        B, _, h, w = z.shape
        zp = torch.zeros(B, h, w, 64, dtype=dt, device=dev)                     # plumbing: NHWC, channels padded to 64
        zp[..., :z.shape[1]] = z.permute(0, 2, 3, 1).to(dt)
        wq, bq = _pad_1x1(self.post_quant_conv, ctx)
        t = ctx.gemm(zp.view(B * h * w, 64), wq, bias=bq, descr="vae.post_quant").view(B, h, w, 64)
        d = self.decoder
        x = ctx.conv3x3(t, _pad_conv_in(d.conv_in, ctx), bias=_b(d.conv_in, ctx), descr="vae.conv_in")
        ctx.free(t)
        x = d.mid_block.emit(ctx, x)
        for blk in d.up_blocks:
            x = blk.emit(ctx, x)
        n = _gn(ctx, d.conv_norm_out, x, d.groups, True, "vae.conv_norm_out")
        ctx.free(x)
        y = ctx.conv3x3(n, d.conv_out.packed(ctx), bias=_b(d.conv_out, ctx), descr="vae.conv_out")    # [B, 8h, 8w, 3]
        ctx.free(n)
        return y.permute(0, 3, 1, 2).float()

    @staticmethod
    def _blend(a, b, extent, dim):
        """diffusers blend_v (dim 2) / blend_h (dim 3): linear cross-fade of b's first `extent` rows with a's last"""
        extent = min(a.shape[dim], b.shape[dim], extent)
        wgt = (torch.arange(extent, device=b.device, dtype=b.dtype) / extent).view([-1 if i == dim else 1 for i in range(4)])
        head = a.narrow(dim, a.shape[dim] - extent, extent) * (1 - wgt) + b.narrow(dim, 0, extent) * wgt
        return torch.cat([head, b.narrow(dim, extent, b.shape[dim] - extent)], dim)

    def tiled_decode(self, z, precision="native"):
        overlap = int(self.tile_latent_min_size * (1 - self.tile_overlap_factor))
        extent = int(self.tile_sample_min_size * self.tile_overlap_factor)
        limit = self.tile_sample_min_size - extent
        # every operation of the decoder is per sample, so the tiles of one shape are decoded as ONE batch (round 6: 3 x 3 tiles of a 128 x 128
        # latent = 4 launches' worth of work instead of 9 under-filled ones; in the fp32 path the same bits per tile as one tile at a time, in the
        # 16-bit path the same to rounding -- there the GEMM variant depends on M)
        ii, jj = list(range(0, z.shape[2], overlap)), list(range(0, z.shape[3], overlap))
        tiles = {(i, j): z[:, :, i:i + self.tile_latent_min_size, j:j + self.tile_latent_min_size] for i in ii for j in jj}
        groups = {}
        for key, t in tiles.items():
            groups.setdefault(tuple(t.shape[2:]), []).append(key)
        dec = {}
        for keys in groups.values():
            out = self._decode_tile(torch.cat([tiles[k] for k in keys], 0).contiguous(), precision)
            for k, o in zip(keys, out.split(z.shape[0], 0)):
                dec[k] = o
        rows = [[dec[(i, j)] for j in jj] for i in ii]
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, t in enumerate(row):
                if i > 0:
                    t = self._blend(rows[i - 1][j], t, extent, 2)
                if j > 0:
                    t = self._blend(row[j - 1], t, extent, 3)
                row[j] = t               # diffusers blends in place: later tiles see the blended neighbour
                out.append(t[:, :, :limit, :limit])
            out_rows.append(torch.cat(out, dim=3))
        return torch.cat(out_rows, dim=2)

    @torch.no_grad()
    def decode(self, z, precision=None):
        """precision: None / 'auto' = the reference's rule (fp32 for a float16 module with force_upcast or a float32 module, the module's
        16-bit dtype for bfloat16); 'fp32' / 'native' force a mode"""
        precision = self.precision_for(precision)
        z = z.to(self.decoder.conv_in.weight.device, torch.float32)
        if self.use_tiling and (z.shape[-1] > self.tile_latent_min_size or z.shape[-2] > self.tile_latent_min_size):
            return self.tiled_decode(z, precision)
        return self._decode_tile(z, precision)


def decode_latents(vae: AutoencoderKL, latents, precision=None):
    """custom_pipelines.py:365-379"""
    return vae.decode(latents.float() / vae.config.scaling_factor, precision=precision)


def postprocess(image, output_type="pil"):
    """VaeImageProcessor.postprocess (do_normalize=True), custom_pipelines.py:386: 'pt' | 'np' | 'pil'"""
    x = (image / 2 + 0.5).clamp(0, 1)
    if output_type == "pt":
        return x
    arr = x.cpu().permute(0, 2, 3, 1).float().numpy()
    if output_type == "np":
        return arr
    if output_type != "pil":
        raise ValueError(f"output_type {output_type!r} (expected 'latent', 'pt', 'np' or 'pil')")
    from PIL import Image
    return [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
