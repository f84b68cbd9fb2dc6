"""GPU parity of the HIP VAE decode (imagharmony_amd.vae) against the CPU fp32 oracle (oracle.vae, a restatement of
diffusers' AutoencoderKL: parity unpinned upstream, structurally pinned by the parameter count) on a reduced-width
config with identical seeded weights.  Tolerance: rel-RMS of the decoded image <= 3e-2 in bf16 (the path the
pipeline uses; ~25 conv / norm layers), 8e-3 in fp16 on the small config (the real SDXL VAE overflows fp16)."""
import pytest
import torch

from conftest import rel_rms
from oracle.detfill import det_fill, det_randn
from oracle.vae import AutoencoderKL as OracleVAE
from oracle.vae import decode_latents as oracle_decode
from oracle.vae import postprocess as oracle_post
from oracle.vae import tiny_vae_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_pair(dtype):
    from imagharmony_amd.vae import AutoencoderKL, VAEConfig
    ocfg = tiny_vae_config()
    ov = det_fill(OracleVAE(ocfg), 3).eval()
    hv = AutoencoderKL(VAEConfig(**{k: getattr(ocfg, k) for k in VAEConfig.__dataclass_fields__}))
    hv.load_state_dict(ov.state_dict(), strict=True)          # encoder / quant_conv keys are accepted and dropped
    return ov, hv.to(DEV, dtype)


def test_row_softmax_op():
    from imagharmony_amd import lib as L
    from imagharmony_amd.ctx import Ctx
    for dtype in (torch.bfloat16, torch.float16):
        ctx = Ctx(DEV, dtype)
        a = (det_randn((130, 1024), 2) * 3.0).to(DEV)
        a[5, 7] = 40.0                                                   # one dominant score
        y = torch.empty(130, 1024, device=DEV, dtype=dtype)
        ctx.ew(L.EW_SOFTMAX, y, a=a, i=(130, 1024, 1024, 1024, 0, 0), f=(0.25, 0.0, 0.0, 0.0))
        ref = torch.softmax(a * 0.25, dim=-1)
        assert torch.allclose(y.float(), ref, atol=4e-3 if dtype == torch.bfloat16 else 5e-4, rtol=2e-2)
        assert torch.allclose(y.float().sum(-1), torch.ones(130, device=DEV), atol=2e-2)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 3e-2), (torch.float16, 8e-3)])
def test_vae_decode_matches_oracle(dtype, tol):
    ov, hv = build_pair(dtype)
    lat = det_randn((2, 4, 32, 32), 5) * 0.13025 * 3.0
    with torch.no_grad():
        ref = oracle_decode(ov, lat)
    from imagharmony_amd.vae import decode_latents, postprocess
    img = decode_latents(hv, lat.to(DEV), precision="native")      # the module's own 16-bit dtype (auto would upcast the fp16 module like the reference)
    assert img.shape == ref.shape == (2, 3, 256, 256) and img.dtype == torch.float32
    r = rel_rms(img.cpu(), ref)
    print(f"vae tiny {dtype}: rel-rms {r:.3e}")
    assert r < tol, r
    a, b = postprocess(img, "np"), oracle_post(ref, "np")
    assert a.shape == b.shape == (2, 256, 256, 3) and abs(a - b).mean() < 2e-2
    assert postprocess(img, "pil")[0].size == (256, 256)


def test_f32_ops_against_torch():
    """csrc/f32.hip, the reference-precision kernels of the decode tail: GEMM (ragged M / N, N = 3, strided W), conv3x3 (+ fused nearest x2),
    GroupNorm (+ SiLU, rows with |mean| >> sigma) and row softmax against fp32 / fp64 torch"""
    import torch.nn.functional as F
    from imagharmony_amd.ctx import Ctx
    ctx = Ctx(DEV, torch.bfloat16)
    # GEMM / conv in both arithmetic modes: imh_debug_set(10, 1) = exact fp32 MFMA for every launch; 0 (default) = operands split into bf16
    # hi + lo, three bf16 MFMAs per product (16 mantissa bits per operand: ~2e-5 per product, random sign) where the K tile fits -- bounds x4
    for exact in (1, 0):
        ctx.lib.imh_debug_set(10, exact)
        tol = 1.0 if exact else 4.0
        try:
            _f32_gemm_conv_cases(ctx, tol)
        finally:
            ctx.lib.imh_debug_set(10, 0)
    _f32_rest(ctx)


def _f32_gemm_conv_cases(ctx, tol):
    import torch.nn.functional as F
    for (M, N, K) in [(300, 200, 64), (128, 3, 1152), (1000, 129, 16), (64, 512, 512)]:
        x, w = det_randn((M, K), 1).to(DEV), det_randn((N, K), 2).to(DEV)
        b, r = det_randn((N,), 3).to(DEV), det_randn((M, N), 4).to(DEV)
        y = ctx.f32_gemm(x, w, bias=b, residual=r)
        ref = (x.double() @ w.double().t() + b.double() + r.double())
        assert (y.double() - ref).abs().max() < tol * 2e-5 * K ** 0.5 * 4, (M, N, K, tol)
    # strided weight operand (the PV GEMM reads V^T [C, B L] one batch at a time)
    vt = det_randn((32, 3 * 128), 5).to(DEV)
    pr = det_randn((128, 128), 6).to(DEV)
    y = ctx.f32_gemm(pr, vt[:, 128:256], N=32, K=128, ldw=384)
    assert (y.double() - pr.double() @ vt[:, 128:256].double().t()).abs().max() < tol * 1e-3
    for (B, H, W, Cin, Cout, up) in [(2, 9, 7, 16, 40, 0), (1, 8, 8, 32, 3, 1), (1, 16, 12, 64, 130, 0)]:
        x = det_randn((B, Cin, H, W), 7).to(DEV)
        w = (det_randn((Cout, Cin, 3, 3), 8) * (9 * Cin) ** -0.5).to(DEV)
        b = det_randn((Cout,), 9).to(DEV)
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
        ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
        res = det_randn(tuple(ref.permute(0, 2, 3, 1).shape), 10).to(DEV)
        y = ctx.f32_conv3x3(x.permute(0, 2, 3, 1).contiguous(), w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous(), bias=b,
                            residual=res.view(-1, Cout), up=up)
        assert (y.double() - (ref.permute(0, 2, 3, 1) + res.double())).abs().max() < tol * 5e-5, (B, H, W, Cin, Cout, up, tol)


def _f32_rest(ctx):
    import torch.nn.functional as F
    for (B, HW, Cc, silu, off) in [(2, 5000, 128, True, 0.0), (1, 1024, 512, False, 300.0), (3, 70, 32, True, -40.0)]:
        x = (det_randn((B, HW, Cc), 11) * 0.7 + off).to(DEV)
        g, be = (1 + 0.2 * det_randn((Cc,), 12)).to(DEV), (0.3 * det_randn((Cc,), 13)).to(DEV)
        y = ctx.f32_groupnorm(x, g, be, 32, 1e-6, silu=silu)
        ref = F.group_norm(x.double().permute(0, 2, 1), 32, g.double(), be.double(), 1e-6).permute(0, 2, 1)
        if silu:
            ref = F.silu(ref)
        assert (y.double() - ref).abs().max() < (2e-5 if off == 0 else 2e-3), (B, HW, Cc, off, (y.double() - ref).abs().max().item())
    a = (det_randn((130, 1000), 14) * 30.0).to(DEV)
    out = torch.empty_like(a)
    ctx.f32_softmax(a, out, 0.25)
    assert (out.double() - torch.softmax(a.double() * 0.25, -1)).abs().max() < 2e-6
    ctx.f32_softmax(a, a, 0.25)                                     # in place, as the decode uses it
    assert torch.equal(a, out)


def _rounded_pair(dtype, scale_conv_in=1.0):
    """oracle and product VAE with IDENTICAL weights: the seeded weights rounded through `dtype` (what a checkpoint stored in that dtype
    holds, and what the reference's upcast_vae() turns back into fp32)"""
    from imagharmony_amd.vae import AutoencoderKL, VAEConfig
    ocfg = tiny_vae_config()
    ov = det_fill(OracleVAE(ocfg), 3).eval()
    with torch.no_grad():
        ov.decoder.conv_in.weight.mul_(scale_conv_in); ov.decoder.conv_in.bias.mul_(scale_conv_in)
        for prm in ov.parameters():
            prm.copy_(prm.to(dtype).float())
    hv = AutoencoderKL(VAEConfig(**{k: getattr(ocfg, k) for k in VAEConfig.__dataclass_fields__}))
    hv.load_state_dict(ov.state_dict(), strict=True)
    return ov, hv.to(DEV, dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_vae_decode_fp32_mode_matches_oracle(dtype):
    """the decode at the REFERENCE's precision (custom_pipelines.py:366-372 upcasts a float16 VAE to fp32 before
    vae.decode): a float16 module (auto -> fp32, like needs_upcasting) and a float32 module decode on csrc/f32.hip with fp32
    activations / weights / arithmetic; rel-RMS vs the fp32 CPU oracle with the same weights <= 2e-3 (measured ~1e-6), untiled and tiled"""
    ov, hv = _rounded_pair(dtype)
    assert hv.precision_for() == "fp32"
    lat = det_randn((2, 4, 32, 32), 5) * 0.13025 * 3.0
    with torch.no_grad():
        ref = oracle_decode(ov, lat)
    from imagharmony_amd.vae import decode_latents
    img = decode_latents(hv, lat.to(DEV))
    assert img.shape == ref.shape == (2, 3, 256, 256) and img.dtype == torch.float32
    r = rel_rms(img.cpu(), ref)
    print(f"vae tiny fp32 mode ({dtype} module): rel-rms {r:.3e}")
    assert r < 2e-3, r
    assert r < 1e-4, r                                               # (what fp32 arithmetic actually delivers)
    if dtype == torch.float16:
        lat2 = det_randn((1, 4, 64, 48), 9) * 0.13025 * 3.0
        ov.enable_tiling(); hv.enable_tiling()
        with torch.no_grad():
            ref2 = oracle_decode(ov, lat2)
        assert rel_rms(decode_latents(hv, lat2.to(DEV)).cpu(), ref2) < 1e-4
        assert hv.to(torch.bfloat16).precision_for() == "native"     # a bfloat16 module is not upcast (nor is it upstream)


def test_vae_fp32_mode_survives_activations_that_overflow_fp16():
    """why the reference upcasts: with conv_in scaled so that the residual stream reaches ~1e5 the float16 decode is not finite, while the
    fp32 mode -- the default for that same float16 module -- still matches the fp32 oracle"""
    ov, hv = _rounded_pair(torch.float16, scale_conv_in=4.0e3)
    lat = det_randn((1, 4, 32, 32), 5) * 0.13025 * 3.0
    with torch.no_grad():
        ref = oracle_decode(ov, lat)
    assert torch.isfinite(ref).all()
    from imagharmony_amd.vae import decode_latents
    native = decode_latents(hv, lat.to(DEV), precision="native")
    assert not torch.isfinite(native).all(), "the float16 decode was expected to overflow on this input"
    img = decode_latents(hv, lat.to(DEV))
    assert torch.isfinite(img).all()
    r = rel_rms(img.cpu(), ref)
    print(f"vae tiny fp32 mode, overflowing residual stream: rel-rms {r:.3e}")
    assert r < 2e-3, r


def test_vae_tiled_decode_matches_oracle_tiled():
    """diffusers' tiled_decode (test.py:73 enables it): overlapping tiles, in-place linear blends"""
    dtype = torch.bfloat16
    ov, hv = build_pair(dtype)
    lat = det_randn((1, 4, 64, 48), 9) * 0.13025 * 3.0            # tiles of 32 latent pixels, 24 apart: 3 x 2 tiles
    ov.enable_tiling(); hv.enable_tiling()
    with torch.no_grad():
        ref = oracle_decode(ov, lat)
    from imagharmony_amd.vae import decode_latents
    img = decode_latents(hv, lat.to(DEV))
    assert img.shape == ref.shape == (1, 3, 512, 384)
    assert rel_rms(img.cpu(), ref) < 3e-2
    hv.enable_tiling(False)
    assert rel_rms(decode_latents(hv, lat.to(DEV)).cpu(), ref) > 1e-3      # tiling really changes the result (per-tile GroupNorm)
    # round 6: the tiles of one shape travel through the decoder as ONE batch -- every operation is per sample, so in the fp32 path (one
    # kernel per op whatever M) each tile's image is the same bits as that tile decoded alone; in the 16-bit path the GEMM variant is
    # chosen by M (split-K, tile shape), so the tile agrees to rounding only
    for mdt, prec in ((torch.bfloat16, "native"), (torch.float16, "fp32")):
        _, hv2 = build_pair(mdt)
        z = (det_randn((2, 4, 64, 48), 10) * 0.4).to(DEV)
        overlap = int(hv2.tile_latent_min_size * (1 - hv2.tile_overlap_factor))
        T = hv2.tile_latent_min_size
        shapes = {}
        for i in range(0, 64, overlap):
            for j in range(0, 48, overlap):
                shapes.setdefault(tuple(z[:, :, i:i + T, j:j + T].shape[2:]), []).append((i, j))
        assert max(len(v) for v in shapes.values()) >= 2, "the case must batch something"
        for keys in shapes.values():
            both = hv2._decode_tile(torch.cat([z[:, :, i:i + T, j:j + T] for (i, j) in keys], 0).contiguous(), prec)
            for k, (i, j) in enumerate(keys):
                alone = hv2._decode_tile(z[:, :, i:i + T, j:j + T].contiguous(), prec)
                if prec == "fp32":
                    assert torch.equal(both[2 * k:2 * k + 2], alone), (mdt, prec, i, j)
                else:
                    assert rel_rms(both[2 * k:2 * k + 2], alone) < 1.5e-2, (mdt, prec, i, j)


def test_pipeline_output_types_with_vae():
    """output_type 'pil' / 'np' / 'pt' through StableDiffusionXLCustomPipeline (custom_pipelines.py:365-386)"""
    from smoke_impl import build_pair as unet_pair
    from imagharmony_amd import schedulers as hs
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    dtype = torch.bfloat16
    ou, hu, ocfg = unet_pair(DEV, dtype)
    ov, hv = build_pair(dtype)
    pipe = StableDiffusionXLCustomPipeline(hu, scheduler=hs.DDIMScheduler(), device=DEV, dtype=dtype, vae=hv)
    pipe.enable_vae_tiling()
    cd = ocfg.cross_attention_dim
    kw = dict(prompt_embeds=det_randn((1, 81, cd), 4), negative_prompt_embeds=det_randn((1, 81, cd), 5),
              pooled_prompt_embeds=det_randn((1, ocfg.pooled_dim), 6), negative_pooled_prompt_embeds=det_randn((1, ocfg.pooled_dim), 7),
              height=256, width=256, num_inference_steps=2, guidance_scale=5.0, latents=det_randn((1, 4, 32, 32), 3))
    lat = pipe(output_type="latent", **kw).images
    pil = pipe(output_type="pil", **kw).images
    arr = pipe(output_type="np", **kw).images
    pt = pipe(output_type="pt", **kw).images
    assert lat.shape == (1, 4, 32, 32) and pil[0].size == (256, 256) and arr.shape == (1, 256, 256, 3)
    assert pt.shape == (1, 3, 256, 256) and float(pt.min()) >= 0.0 and float(pt.max()) <= 1.0
    with torch.no_grad():
        ref = oracle_post(oracle_decode(ov, lat.cpu()), "np")
    assert abs(arr - ref).mean() < 2e-2
