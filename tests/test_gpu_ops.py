"""GPU parity of every HIP kernel behind the C ABI against a plain fp32 PyTorch statement of the
same op (run with ``pytest -m gpu`` on an MI355X).  Tolerances: outputs are bf16/fp16 with fp32
accumulation; the bound used is a few output-dtype ulps of the result scale (stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import built, experimental

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.bfloat16, torch.float16]
EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


@pytest.fixture(scope="module")
def L():
    from imagharmony_amd import lib
    lib.load()
    return lib


def ctx_for(dtype):
    from imagharmony_amd.ctx import Ctx
    return Ctx(DEV, dtype)


def rnd(*shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def geglu_ref(full):
    """GF_GEGLU on pre-activations whose columns are interleaved in (value, value, gate, gate) quads"""
    q = full.reshape(full.shape[0], -1, 4)
    return (q[:, :, :2] * F.gelu(q[:, :, 2:])).reshape(full.shape[0], -1)


def assert_close(y, ref, dtype, what, k=4.0):
    ref = ref.float()
    y = y.float()
    scale = ref.abs().max().item() + 1e-6
    err = (y - ref).abs().max().item()
    rms = ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12)).item()
    assert math.isfinite(err), f"{what}: non-finite output"
    assert err <= k * EPS[dtype] * scale and rms <= 2 * EPS[dtype], f"{what}: max err {err:.3e} (scale {scale:.3e}), rel-rms {rms:.3e}"


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(128, 128, 1), (128, 64, 1), (64, 128, 1), (64, 64, 1), (128, 128, 2), (64, 64, 4)])
@pytest.mark.parametrize("shape", [(256, 256, 128), (300, 200, 192), (64, 1280, 640), (2, 320, 64), (130, 72, 64)])
def test_gemm_plain(L, dtype, cfg, shape):
    M, N, K = shape
    ctx = ctx_for(dtype)
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    y = ctx.gemm(x, w, cfg=cfg)
    assert_close(y, x.float() @ w.float().t(), dtype, f"gemm {shape} {cfg}")


# variant codes of the big-tile / ring / ping-pong kernels (gemm_ring.hip, gemm_pp.hip): (bm code, bn, splits)
BIG_VARIANTS = [(256, 128, 1), (256, 256, 1), (3128, 128, 1), (3064, 64, 1), (4128, 64, 1), (5064, 64, 1), (4064, 64, 1),
                (4064, 128, 1), (6128, 320, 1), (5258, 320, 1), (6064, 160, 1), (8256, 256, 1), (9128, 320, 1), (9256, 320, 1),
                (1464, 160, 1), (2464, 160, 1), (2464, 160, 2), (24128, 160, 1), (24128, 128, 1), (23256, 160, 1), (23256, 128, 1), (22128, 160, 1)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", built(BIG_VARIANTS))
def test_gemm_big_tile_variants(L, dtype, cfg):
    """every ring / KG2 / ping-pong variant on tile-aligned, ragged and single-K-tile shapes, with bias + residual, bitwise
    repeatable (the counted-vmcnt pipelines are race-screened by repetition)"""
    ctx = ctx_for(dtype)
    for (M, N, K) in [(512, 640, 64), (300, 520, 192), (640, 1280, 320), (2, 320, 128), (128, 320, 128), (192, 160, 576)]:
        x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
        b, r = rnd(N, dtype=dtype, seed=3), rnd(M, N, dtype=dtype, seed=4)
        ref = x.float() @ w.float().t() + b.float() + r.float()
        y0 = ctx.gemm(x, w, bias=b, residual=r, cfg=cfg).clone()
        assert_close(y0, ref, dtype, f"gemm {cfg} {(M, N, K)}")
        for _ in range(3):
            assert torch.equal(ctx.gemm(x, w, bias=b, residual=r, cfg=cfg), y0), f"{cfg} {(M, N, K)} not repeatable"
        # the XCD cell shape of the tile grid (imh_gemm_args.xcd) is placement only: every tile is computed exactly once whatever the shape
        try:
            for cells in (2, 3, 4, 5):
                ctx.xcd_cells = cells
                assert torch.equal(ctx.gemm(x, w, bias=b, residual=r, cfg=cfg), y0), f"{cfg} {(M, N, K)} xcd cells {cells}"
        finally:
            ctx.xcd_cells = 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", built([(23256, 160, 1), (23256, 128, 1), (24128, 160, 1), (24128, 128, 1), (2464, 160, 1), (1464, 160, 1), (22128, 160, 1)]))
def test_gemm_wave_specialised_folded_layernorm_geglu(L, dtype, cfg):
    """the wave-specialised kernels with the folded LayerNorm (row statistics from the row-statistics kernel, supplied by
    Ctx.gemm) and the GEGLU epilogue -- the ff.net.0 launch -- against F.layer_norm + matmul + gelu in fp32; the ping-pong
    variants, whose folded form took its statistics inside the K loop, refuse the flag now"""
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    for (M, N, K) in [(512, 640, 128), (300, 960, 320), (2048, 2560, 640)]:
        x = (rnd(M, K, dtype=dtype, seed=1) * 1.5 + 2.0).contiguous()
        w = rnd(N, K, dtype=torch.float32, seed=2, scale=K ** -0.5)
        norm = torch.nn.LayerNorm(K, eps=1e-5)
        with torch.no_grad():
            norm.weight.copy_(1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3)))
            norm.bias.copy_(0.3 * torch.randn(K, generator=torch.Generator().manual_seed(4)))
        full = (F.layer_norm(x.float().cpu(), (K,), norm.weight, norm.bias, 1e-5) @ w.cpu().t()).to(DEV)
        wg, s, c = fold_ln(w, norm, ctx)
        y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5), cfg=cfg)
        assert_close(y, full, dtype, f"folded LN {cfg} {(M, N, K)}", k=6.0)
        g = ctx.gemm(x, wg, flags=L.GF_LN_ROW | L.GF_GEGLU, ln=(s, c, 1e-5), cfg=cfg)
        assert_close(g, geglu_ref(full), dtype, f"folded LN + GEGLU {cfg} {(M, N, K)}", k=8.0)
        assert torch.equal(g, ctx.gemm(x, wg, flags=L.GF_LN_ROW | L.GF_GEGLU, ln=(s, c, 1e-5), cfg=cfg))
    for pp in built([(8256, 256, 1), (9128, 320, 1), (9256, 320, 1)]):
        with pytest.raises(L.ImhError, match="folded LayerNorm"):
            ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5), cfg=pp)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2464, 160, 1), (1464, 160, 1), (24128, 160, 1), (23256, 160, 1), (128, 128, 1)])
def test_residual_add_in_place_and_early_fetch(L, dtype, cfg):
    """to_out / ff.out / proj_out (attention_processor.py:320-329, 453-462: linear + bias, then + residual) on the wave-specialised
    kernels, whose residual rows and bias are fetched BEFORE the K loop (round 5): same bits as the late fetch (imh_debug_set(6, 0)),
    also IN PLACE (Y == residual: every element is read by the lane that later stores it), ragged M, statistics epilogue unchanged"""
    ctx = ctx_for(dtype)
    for (M, N, K) in [(2048, 1280, 1280), (300, 640, 256), (8192, 640, 2560)]:
        x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
        b, r = rnd(N, dtype=dtype, seed=3), rnd(M, N, dtype=dtype, seed=4)
        ref = x.float() @ w.float().t() + b.float() + r.float()
        y, st = ctx.gemm(x, w, bias=b, residual=r, cfg=cfg, stats_out=True)
        assert_close(y, ref, dtype, f"bias + residual {cfg} {(M, N, K)}")
        try:
            assert L.load().imh_debug_set(6, 0) == 0
            y0, st0 = ctx.gemm(x, w, bias=b, residual=r, cfg=cfg, stats_out=True)
        finally:
            L.load().imh_debug_set(6, 1)
        assert torch.equal(y, y0) and torch.equal(st[0], st0[0]), "early and late residual fetch give the same bits"
        rin = r.clone()
        yin = ctx.gemm(x, w, bias=b, residual=rin, out=rin, cfg=cfg)
        assert yin.data_ptr() == rin.data_ptr() and torch.equal(yin, y), "in place"


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(L, dtype):
    ctx = ctx_for(dtype)
    M, N, K = 192, 256, 128
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b, r = rnd(N, dtype=dtype, seed=3), rnd(M, N, dtype=dtype, seed=4)
    ref = x.float() @ w.float().t()
    assert_close(ctx.gemm(x, w, bias=b), ref + b.float(), dtype, "bias")
    assert_close(ctx.gemm(x, w, bias=b, residual=r), ref + b.float() + r.float(), dtype, "bias+residual")
    assert_close(ctx.gemm(x, w, bias=b, flags=L.GF_ACT_SILU), F.silu(ref + b.float()), dtype, "silu")
    assert_close(ctx.gemm(x, w, flags=L.GF_ACT_GELU), F.gelu(ref), dtype, "gelu")
    # large-magnitude pre-activations (beyond the polynomial's fit range |x| <= 4.67): GELU is exactly x or 0 there, not x * (1 +- 3e-6)
    big = ctx.gemm((x * 64).to(dtype), w, flags=L.GF_ACT_GELU | L.GF_OUT_F32).float()
    ref_big = (x * 64).to(dtype).float() @ w.float().t()
    far = ref_big.abs() > 6.0
    assert far.float().mean() > 0.5
    assert torch.equal(big[far & (ref_big < 0)], torch.zeros_like(big[far & (ref_big < 0)]))
    assert (big[far & (ref_big > 0)] - ref_big[far & (ref_big > 0)]).abs().max() <= 1e-4 * ref_big.abs().max()
    # rowadd: 3 batches of 64 rows, row stride larger than N (stacked time_emb_proj layout)
    ra_full = rnd(3, N + 64, dtype=dtype, seed=5)
    ra = ra_full[:, 32:32 + N]
    y = ctx.gemm(x, w, bias=b, rowadd=ra, rows_per_batch=64, ldra=ra_full.stride(0))
    assert_close(y, ref + b.float() + ra.float().repeat_interleave(64, 0), dtype, "rowadd")
    # split-K with every epilogue
    y = ctx.gemm(x, w, bias=b, residual=r, cfg=(64, 64, 2))
    assert_close(y, ref + b.float() + r.float(), dtype, "splitk epilogue")
    # fp32 output
    y = ctx.gemm(x, w, flags=L.GF_OUT_F32)
    assert y.dtype == torch.float32
    assert_close(y, ref, dtype, "f32 out")
    # strided operands (column slices)
    xb = rnd(M, K + 64, dtype=dtype, seed=6)
    y = ctx.gemm(xb[:, 64:], w)
    assert_close(y, xb[:, 64:].float() @ w.float().t(), dtype, "strided x")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(128, 128, 1), (64, 64, 1), (128, 64, 2)])
def test_gemm_geglu(L, dtype, cfg):
    ctx = ctx_for(dtype)
    M, K, inner = 160, 128, 256
    x = rnd(M, K, dtype=dtype, seed=1)
    w, b = rnd(2 * inner, K, dtype=dtype, seed=2, scale=K ** -0.5), rnd(2 * inner, dtype=dtype, seed=3)
    r = rnd(M, inner, dtype=dtype, seed=4)
    from imagharmony_amd.unet import geglu_interleave
    wi, bi = geglu_interleave(w), geglu_interleave(b)
    y = ctx.gemm(x, wi, bias=bi, flags=L.GF_GEGLU, residual=r, cfg=cfg)
    full = x.float() @ w.float().t() + b.float()
    ref = full[:, :inner] * F.gelu(full[:, inner:]) + r.float()
    assert y.shape == (M, inner)
    assert_close(y, ref, dtype, f"geglu {cfg}")


def vt_unpermute(vt):
    """[C, n] with 16-groups stored as [0-3, 8-11, 4-7, 12-15] -> logical order"""
    C_, n = vt.shape
    v = vt.view(C_, n // 16, 4, 4)
    return v[:, :, [0, 2, 1, 3], :].reshape(C_, n)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(128, 128, 1), (64, 128, 1), (64, 64, 2)])
def test_gemm_vt_perm(L, dtype, cfg):
    ctx = ctx_for(dtype)
    C_, n, K = 128, 192, 64
    wv, x = rnd(C_, K, dtype=dtype, seed=1, scale=K ** -0.5), rnd(n, K, dtype=dtype, seed=2)
    vt = ctx.gemm(wv, x, flags=L.GF_VT_PERM, cfg=cfg)
    assert_close(vt_unpermute(vt), wv.float() @ x.float().t(), dtype, "V^T")


# ------------------------------------------------------------------------------------ conv
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", built([dict(B=2, H=16, W=16, Cin=64, Cout=128), dict(B=1, H=12, W=20, Cin=128, Cout=64),
                                  dict(B=2, H=16, W=16, Cin=64, Cout=64, stride=2), dict(B=2, H=8, W=8, Cin=64, Cout=64, up=1),
                                  dict(B=2, H=8, W=8, Cin=192, Cout=4), dict(B=1, H=32, W=32, Cin=320, Cout=320, cfg=(128, 128, 2)),
                                  dict(B=1, H=32, W=32, Cin=128, Cout=640, cfg=(5258, 320, 1)), dict(B=2, H=16, W=24, Cin=64, Cout=320, up=1, cfg=(6128, 320, 1)),
                                  dict(B=1, H=24, W=24, Cin=64, Cout=128, stride=2, cfg=(4128, 64, 1)),
                                  # halo kernel: aligned, ragged patch grid, several cout tiles, fused x2 upsampling, tiny
                                  dict(B=2, H=16, W=32, Cin=128, Cout=320, cfg=(7128, 320, 1)), dict(B=1, H=12, W=20, Cin=64, Cout=64, cfg=(7128, 320, 1)),
                                  dict(B=1, H=24, W=16, Cin=192, Cout=704, cfg=(7128, 320, 1)), dict(B=2, H=8, W=8, Cin=64, Cout=320, up=1, cfg=(7128, 320, 1)),
                                  dict(B=1, H=3, W=5, Cin=64, Cout=8, cfg=(7128, 320, 1)),
                                  dict(B=2, H=16, W=16, Cin=128, Cout=640, cfg=(7128, 160, 1)), dict(B=1, H=12, W=20, Cin=64, Cout=200, up=1, cfg=(7128, 160, 1)),
                                  dict(B=2, H=16, W=16, Cin=128, Cout=640, cfg=(7328, 160, 1)), dict(B=1, H=12, W=20, Cin=64, Cout=200, up=1, cfg=(7428, 160, 1)),
                                  dict(B=1, H=24, W=16, Cin=192, Cout=320, cfg=(7428, 160, 1)), dict(B=2, H=32, W=32, Cin=320, Cout=640, cfg=(7328, 160, 1)),
                                  dict(B=2, H=32, W=32, Cin=128, Cout=320, cfg=(7256, 160, 1)), dict(B=1, H=20, W=12, Cin=64, Cout=200, up=1, cfg=(7356, 160, 1)),
                                  dict(B=1, H=24, W=40, Cin=192, Cout=160, cfg=(7356, 160, 1)),
                                  dict(B=2, H=16, W=16, Cin=128, Cout=320, cfg=(7564, 160, 1)), dict(B=1, H=10, W=20, Cin=64, Cout=384, cfg=(7564, 320, 1)),
                                  # 8 x 16 patch x 80 couts, the wave pairs split K (round 5): aligned, ragged patch grid + ragged cout tile, fused upsampling
                                  dict(B=2, H=16, W=32, Cin=128, Cout=320, cfg=(7128, 80, 1)), dict(B=1, H=12, W=20, Cin=192, Cout=200, cfg=(7128, 80, 1)),
                                  dict(B=2, H=8, W=8, Cin=64, Cout=160, up=1, cfg=(7128, 80, 1)),
                                  dict(B=2, H=16, W=16, Cin=128, Cout=320, cfg=(1464, 160, 1)), dict(B=1, H=12, W=20, Cin=64, Cout=200, up=1, cfg=(2464, 160, 1)),
                                  dict(B=1, H=24, W=24, Cin=64, Cout=160, stride=2, cfg=(2464, 160, 2)),
                                  dict(B=2, H=16, W=16, Cin=128, Cout=320, cfg=(24128, 160, 1)), dict(B=2, H=16, W=16, Cin=128, Cout=320, cfg=(22128, 160, 1)), dict(B=1, H=12, W=20, Cin=64, Cout=200, up=1, cfg=(24128, 128, 1))]))
def test_conv3x3(L, dtype, case):
    ctx = ctx_for(dtype)
    B, H, W, Cin, Cout = case["B"], case["H"], case["W"], case["Cin"], case["Cout"]
    stride, up = case.get("stride", 1), case.get("up", 0)
    x = rnd(B, H, W, Cin, dtype=dtype, seed=1)
    w = rnd(Cout, Cin, 3, 3, dtype=dtype, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, dtype=dtype, seed=3)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    xin = x.float().permute(0, 3, 1, 2)
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
    temb = rnd(B, Cout, dtype=dtype, seed=4)
    res = rnd(*ref.shape, dtype=dtype, seed=5)
    y = ctx.conv3x3(x, wp, bias=b, stride=stride, up=up, rowadd=temb, residual=res.view(-1, Cout), cfg=case.get("cfg"))
    assert y.shape == ref.shape
    assert_close(y, ref + temb.float()[:, None, None, :] + res.float(), dtype, f"conv {case}")


# ------------------------------------------------------------------------------------ attention
def make_vt(v, n_pad):
    """v [B, n, H*64] -> permuted V^T [H*64, B*n_pad]"""
    B, n, C_ = v.shape
    vp = torch.zeros(B, n_pad, C_, dtype=v.dtype, device=v.device)
    vp[:, :n] = v
    vt = vp.permute(2, 0, 1).reshape(C_, B * n_pad // 16, 4, 4)
    return vt[:, :, [0, 2, 1, 3], :].reshape(C_, B * n_pad).contiguous()


def sdpa_ref(q, k, v, H):
    B, Lq, C_ = q.shape
    hs = lambda t: t.float().view(B, -1, H, 64).transpose(1, 2)
    o = F.scaled_dot_product_attention(hs(q), hs(k), hs(v))
    return o.transpose(1, 2).reshape(B, Lq, C_)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [0, 1, 2, 3] + ([5, 6] if experimental() else []))      # 5 / 6: the key-split kernel (-DIMH_EXPERIMENTAL)
@pytest.mark.parametrize("B,H,Lq", [(2, 2, 256), (1, 5, 1024), (2, 1, 64), (1, 2, 192), (1, 3, 128), (2, 1, 320), (1, 2, 384), (1, 1, 448),
                                    (1, 1, 512), (1, 2, 4096), (2, 3, 640), (1, 2, 768)])
def test_attention_self(L, dtype, mode, B, H, Lq):
    """mode (imh_debug_set key 4): 0 = what the forward runs, 1 = in-order key loop, 2 / 3 = software-pipelined key loop with the
    textbook / the deferred running maximum (every tail of its unrolled tile loop: 1 .. 8, 16, 64 tiles; a key count that is not a
    multiple of 64 always takes the in-order kernel), 5 / 6 = key-split workgroups (two key halves x four query groups, merged
    in LDS) with the deferred / the textbook maximum: an even number (>= 4) of whole key tiles, else the next kernel in line"""
    assert L.load().imh_debug_set(4, mode) == 0
    try:
        _attention_self_case(L, dtype, B, H, Lq)
    finally:
        L.load().imh_debug_set(4, 0)


def _attention_self_case(L, dtype, B, H, Lq):
    ctx = ctx_for(dtype)
    C_ = H * 64
    qk = rnd(B * Lq, 2 * C_, dtype=dtype, seed=1)
    v = rnd(B, Lq, C_, dtype=dtype, seed=2)
    vt = make_vt(v, Lq)
    out = ctx.new(B * Lq, C_)
    ctx.attention(qk[:, :C_], qk[:, C_:], vt, out, B, H, Lq, Lq, Lq, 2 * C_, 2 * C_, B * Lq, C_, 0.125)
    ref = sdpa_ref(qk[:, :C_].reshape(B, Lq, C_), qk[:, C_:].reshape(B, Lq, C_), v, H)
    assert_close(out.view(B, Lq, C_), ref, dtype, "self attention", k=6.0)
    first = out.clone()
    for _ in range(3):                                      # race screen of the rings / the key-split merge: bitwise repeatable
        out.zero_()
        ctx.attention(qk[:, :C_], qk[:, C_:], vt, out, B, H, Lq, Lq, Lq, 2 * C_, 2 * C_, B * Lq, C_, 0.125)
        assert torch.equal(out, first), "self attention not bitwise repeatable"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Lq,nt,nip", [(2, 2, 256, 77, 4), (1, 20, 128, 77, 16), (2, 1, 100, 77, 32), (1, 2, 64, 130, 0)])
def test_attention_cross_ip(L, dtype, B, H, Lq, nt, nip):
    ctx = ctx_for(dtype)
    C_ = H * 64
    q = rnd(B * Lq, C_, dtype=dtype, seed=1)
    k, v = rnd(B, nt, C_, dtype=dtype, seed=2), rnd(B, nt, C_, dtype=dtype, seed=3)
    pad = lambda n: (n + 63) // 64 * 64
    kp = torch.zeros(B, pad(nt), C_, dtype=dtype, device=DEV)
    kp[:, :nt] = k
    out = ctx.new(B * Lq, C_)
    ref = sdpa_ref(q.view(B, Lq, C_), k, v, H)
    kw = {}
    if nip:
        k2, v2 = rnd(B, nip, C_, dtype=dtype, seed=4), rnd(B, nip, C_, dtype=dtype, seed=5)
        k2p = torch.zeros(B, pad(nip), C_, dtype=dtype, device=DEV)
        k2p[:, :nip] = k2
        kw = dict(k2=k2p, vt2=make_vt(v2, pad(nip)), Lk2=nip, Lk2_pad=pad(nip), ldk2=C_, ldvt2=B * pad(nip), scale2=0.7)
        ref = ref + 0.7 * sdpa_ref(q.view(B, Lq, C_), k2, v2, H)
    ctx.attention(q, kp, make_vt(v, pad(nt)), out, B, H, Lq, nt, pad(nt), C_, C_, B * pad(nt), C_, 0.125, **kw)
    assert_close(out.view(B, Lq, C_), ref, dtype, "cross attention", k=6.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [1] + ([2, 3, 4] if experimental() else []))      # 2 .. 4: the two-head kernel (-DIMH_EXPERIMENTAL)
@pytest.mark.parametrize("B,H,Lq,nt,nip,ln", [(2, 2, 256, 77, 4, 1), (1, 20, 128, 77, 16, 2), (2, 5, 100, 77, 32, 0), (2, 4, 100, 77, 32, 2),
                                               (1, 10, 192, 130, 0, 1), (2, 20, 1024, 77, 4, 2), (2, 20, 1024, 77, 0, 3), (2, 10, 4096, 77, 0, 0)])
def test_fused_cross_attention(L, dtype, mode, B, H, Lq, nt, nip, ln):
    """csrc/xattn.hip -- to_q (+ folded LayerNorm) + text attention (+ image-prompt attention, text + s * ip) in ONE
    launch against the same ops in fp32 torch: q = LN(x) Wq^T rounded to the compute dtype (as attn.to_q does), then
    SDPA per key set.  K caches carry the head dims in the permuted order the kernel's hand-over expects.
    mode (imh_debug_set key 3): 1 = one head per workgroup; 2 / 3 / 4 = two heads per workgroup with 0 / 2 / 4 producer
    waves (an odd head count always takes the one-head kernel).  ln: 0 none, 1 no statistics from the caller (Ctx runs the
    row-statistics kernel), 2 / 3 = handed-over statistics in 32-wide slots / from the row-statistics kernel."""
    from conftest import ref_row_stats
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    assert L.load().imh_debug_set(3, mode) == 0
    try:
        _fused_cross_attention_case(L, ctx, dtype, B, H, Lq, nt, nip, ln, ref_row_stats, fold_ln)
    finally:
        L.load().imh_debug_set(3, 0)


def _fused_cross_attention_case(L, ctx, dtype, B, H, Lq, nt, nip, ln, ref_row_stats, fold_ln):
    C_ = H * 64
    x = (rnd(B * Lq, C_, dtype=dtype, seed=1) * 1.3 + (0.7 if ln else 0.0)).contiguous()
    wq = rnd(C_, C_, dtype=torch.float32, seed=6, scale=C_ ** -0.5)
    k, v = rnd(B, nt, C_, dtype=dtype, seed=2), rnd(B, nt, C_, dtype=dtype, seed=3)
    pad = lambda n: (n + 63) // 64 * 64

    def make_k(kk, n_pad):            # [B, n, C] -> zero-padded, every 16-group of head dims stored as [0-3, 8-11, 4-7, 12-15]
        kp = torch.zeros(B, n_pad, C_, dtype=dtype, device=DEV)
        kp[:, :kk.shape[1]] = kk
        return kp.view(B, n_pad, C_ // 16, 4, 4)[:, :, :, [0, 2, 1, 3], :].reshape(B, n_pad, C_).contiguous()

    if ln:
        norm = torch.nn.LayerNorm(C_, eps=1e-5)
        with torch.no_grad():
            norm.weight.copy_(1 + 0.2 * torch.randn(C_, generator=torch.Generator().manual_seed(3)))
            norm.bias.copy_(0.3 * torch.randn(C_, generator=torch.Generator().manual_seed(4)))
        wg, s_, c_ = fold_ln(wq, norm, ctx)
        st = None if ln == 1 else ((ref_row_stats(x.float(), C_ // 32).to(DEV), C_ // 32) if ln == 2 else ctx.row_stats(x))
        lnq = (s_, c_, 1e-5, st)
        xn = F.layer_norm(x.float(), (C_,), norm.weight.to(DEV), norm.bias.to(DEV), 1e-5)
        q_ref = (xn @ wq.to(DEV).t()).to(dtype)
    else:
        wg, lnq = wq.to(DEV, dtype), None
        q_ref = (x.float() @ wg.float().t()).to(dtype)
    ref = sdpa_ref(q_ref.view(B, Lq, C_), k, v, H)
    kw = {}
    if nip:
        k2, v2 = rnd(B, nip, C_, dtype=dtype, seed=4), rnd(B, nip, C_, dtype=dtype, seed=5)
        kw = dict(k2=make_k(k2, pad(nip)), vt2=make_vt(v2, pad(nip)), Lk2=nip, Lk2_pad=pad(nip), ldk2=C_, ldvt2=B * pad(nip), scale2=0.7)
        ref = ref + 0.7 * sdpa_ref(q_ref.view(B, Lq, C_), k2, v2, H)
    out = ctx.new(B * Lq, C_)
    ctx.cross_attention(x, wg, make_k(k, pad(nt)), make_vt(v, pad(nt)), out, B, H, Lq, nt, pad(nt), C_, B * pad(nt), 0.125,
                        ln=lnq, **kw)
    # the reference rounds q once (bf16 / fp16) exactly like the kernel's hand-over; k = 8 ulps of the output scale
    assert_close(out.view(B, Lq, C_), ref, dtype, f"fused cross attention B={B} H={H} Lq={Lq} nt={nt} nip={nip} ln={ln}", k=8.0)
    return out


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Lq,nt,nip,ln", [(1, 5, 128, 77, 4, 2), (2, 10, 256, 77, 16, 0), (1, 5, 256, 130, 0, 1), (3, 5, 128, 64, 70, 3),
                                               (8, 20, 1024, 77, 16, 2), (8, 20, 1024, 77, 32, 2), (2, 10, 4096, 77, 0, 0)])
def test_fused_cross_attention_wide_form(L, dtype, B, H, Lq, nt, nip, ln):
    """Round 6: the shape-selected WIDE form of csrc/xattn.hip (one workgroup = batch x 128 queries x FIVE heads, producer waves own
    the LDS-DMA rings and stream the K / V^T tiles of the key phase; imh_debug_set(3, 10)) -- (a) against the fp32 torch reference of
    ip_adapter/attention_processor.py:396-450 like the one-head kernel, (b) BIT-identical to the one-head kernel on the same inputs
    (same arithmetic in the same order), (c) bitwise repeatable (race screen of the split X / W rings and the key ring; the 130-key case
    wraps the four-slot key ring three times, the 70 image tokens take two image-prompt tiles).  (8, 20, 1024, 77, 16 | 32) are the
    benchmarked UNet-batch-8 calls of BASELINE.json configs[3] / configs[4], which auto mode routes here."""
    from conftest import ref_row_stats
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    lib = L.load()
    outs = {}
    for mode in (10, 1):
        assert lib.imh_debug_set(3, mode) == 0
        try:
            outs[mode] = _fused_cross_attention_case(L, ctx, dtype, B, H, Lq, nt, nip, ln, ref_row_stats, fold_ln).clone()
            if mode == 10:
                for _ in range(2):
                    again = _fused_cross_attention_case(L, ctx, dtype, B, H, Lq, nt, nip, ln, ref_row_stats, fold_ln)
                    assert torch.equal(again, outs[10]), "wide fused cross attention not bitwise repeatable"
        finally:
            lib.imh_debug_set(3, 0)
    # same arithmetic in the same order: the two-pass (text + image-prompt) instantiations are bit-identical; the text-only fp16 ones differ in
    # the last bit of a few elements (the compiler contracts l = l * alpha + sum differently in the two kernels) -- held to 2 ulp of the output
    d = (outs[10].float() - outs[1].float()).abs()
    ulp = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * outs[1].float().abs().clamp_min(2.0 ** -6)
    assert (d <= 2 * ulp).all(), f"wide form differs from the one-head kernel: max |d| = {d.max().item():.3e}"
    if nip:
        assert torch.equal(outs[10], outs[1]), f"two-pass wide form not bit-identical to the one-head kernel: max |d| = {d.max().item():.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [1] + ([3] if experimental() else []))
@pytest.mark.parametrize("how", ["epilogue", "kernel", "ctx"])
def test_fused_cross_attention_large_mean_rows(L, dtype, mode, how):
    """norm2 folded into the fused cross-attention's to_q on rows with |mean| >> sigma (x = 50 + N(0, 0.1) as
    stored: 400-500 sigma) and one zero-variance row, against F.layer_norm + Linear + SDPA in fp32 -- torch.nn.LayerNorm ahead of
    attn.to_q, ip_adapter/attention_processor.py:396.  Statistics from a producer GEMM's epilogue (what the forward does), from
    the row-statistics kernel, and supplied by Ctx when the caller passes none: the three ways a launch can get them."""
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    B, H, Lq, nt, nip = 2, 20, 256, 77, 4
    C_ = H * 64
    x = (50.0 + 0.1 * rnd(B * Lq, C_, dtype=torch.float32, seed=1)).to(dtype).contiguous()
    x[5] = 50.0
    wq = rnd(C_, C_, dtype=torch.float32, seed=6, scale=C_ ** -0.5)
    k, v = rnd(B, nt, C_, dtype=dtype, seed=2), rnd(B, nt, C_, dtype=dtype, seed=3)
    k2, v2 = rnd(B, nip, C_, dtype=dtype, seed=4), rnd(B, nip, C_, dtype=dtype, seed=5)
    pad = lambda n: (n + 63) // 64 * 64

    def make_k(kk, n_pad):
        kp = torch.zeros(B, n_pad, C_, dtype=dtype, device=DEV)
        kp[:, :kk.shape[1]] = kk
        return kp.view(B, n_pad, C_ // 16, 4, 4)[:, :, :, [0, 2, 1, 3], :].reshape(B, n_pad, C_).contiguous()

    norm = torch.nn.LayerNorm(C_, eps=1e-5)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(C_, generator=torch.Generator().manual_seed(3)))
        norm.bias.copy_(0.3 * torch.randn(C_, generator=torch.Generator().manual_seed(4)))
    wg, s_, c_ = fold_ln(wq, norm, ctx)
    if how == "epilogue":       # y = 0 @ W + residual(x) through the wave-specialised kernel: its epilogue leaves x's statistics
        xx, st = ctx.gemm(torch.zeros(B * Lq, 64, dtype=dtype, device=DEV), torch.zeros(C_, 64, dtype=dtype, device=DEV), residual=x,
                          cfg=(2464, 160, 1), stats_out=True)
        assert torch.equal(xx, x) and st[1] == C_ // 80
    else:
        st = ctx.row_stats(x) if how == "kernel" else None
    xn = F.layer_norm(x.float(), (C_,), norm.weight.to(DEV), norm.bias.to(DEV), 1e-5)
    q32 = xn @ wq.to(DEV).t()
    ref = sdpa_ref(q32.view(B, Lq, C_), k, v, H) + 0.7 * sdpa_ref(q32.view(B, Lq, C_), k2, v2, H)
    assert L.load().imh_debug_set(3, mode) == 0
    try:
        out = ctx.new(B * Lq, C_)
        ctx.cross_attention(x, wg, make_k(k, pad(nt)), make_vt(v, pad(nt)), out, B, H, Lq, nt, pad(nt), C_, B * pad(nt), 0.125,
                            ln=(s_, c_, 1e-5, st), k2=make_k(k2, pad(nip)), vt2=make_vt(v2, pad(nip)), Lk2=nip, Lk2_pad=pad(nip),
                            ldk2=C_, ldvt2=B * pad(nip), scale2=0.7)
    finally:
        L.load().imh_debug_set(3, 0)
    o = out.float().view(B, Lq, C_)
    assert torch.isfinite(o).all()
    # q = rstd * (acc - mean * s) + c: acc and mean * s are ~ 500 sigma of q each and cancel in fp32 (|acc| ~ 50 * |W| sqrt(C)); what is
    # left is noise of a few 1e-2 on q ~ N(0, 1), far below what flips an attention row -- rel-rms of the output, not ulps
    from conftest import rel_rms
    r = rel_rms(o, ref)
    assert r < (0.05 if dtype == torch.bfloat16 else 0.02), f"fused cross attention on large-mean rows ({how}, mode {mode}): rel-rms {r:.3e}"
    # the zero-variance row: x - mean = 0 exactly in torch -> q = beta-term only; the folded form must stay finite and close
    zr = rel_rms(o.view(B * Lq, C_)[5], ref.reshape(B * Lq, C_)[5])
    assert zr < 0.2, f"zero-variance row rel-rms {zr:.3e}"


@pytest.mark.parametrize("mode", [1, 2, 3] + ([5] if experimental() else []))
def test_attention_spiked_scores(L, mode):
    """forces the online-softmax rescale path: one key dominates late in the sequence (every key loop)"""
    assert L.load().imh_debug_set(4, mode) == 0
    try:
        _spiked_case(L)
    finally:
        L.load().imh_debug_set(4, 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [1, 2, 3] + ([5, 6] if experimental() else []))
def test_attention_creeping_maximum(L, dtype, mode):
    """the deferred running maximum (mode 3 = the default of the pipelined loop): every 64-key tile raises the row maxima by
    ~3 in the exponent domain, below the 2^8 deferral threshold per tile but 45 in total -- the rescale must fire every third
    tile or so, P stays <= 2^8, and the result matches the reference like the textbook rule does"""
    assert L.load().imh_debug_set(4, mode) == 0
    try:
        ctx = ctx_for(dtype)
        B, H, Lq = 2, 2, 1024
        g = torch.Generator(device="cpu").manual_seed(5)
        q = torch.randn(B, Lq, H, 64, generator=g)
        k = torch.randn(B, Lq, H, 64, generator=g) * 0.2
        # a common direction u: q . u ~ +8 for every query, k . u grows by 3 / (0.125 * 8 * log2(e)) per tile
        u = torch.nn.functional.normalize(torch.randn(64, generator=g), dim=0)
        q = q + 8.0 * u
        ramp = (torch.arange(Lq) // 64).float() * (3.0 / (0.125 * 8.0 * 1.4427))
        k = k + ramp[None, :, None, None] * u
        qk = torch.cat([q.reshape(B * Lq, H * 64), k.reshape(B * Lq, H * 64)], 1).to(dtype).to(DEV)
        v = rnd(B, Lq, H * 64, dtype=dtype, seed=2)
        out = ctx.new(B * Lq, H * 64)
        C_ = H * 64
        ctx.attention(qk[:, :C_], qk[:, C_:], make_vt(v, Lq), out, B, H, Lq, Lq, Lq, 2 * C_, 2 * C_, B * Lq, C_, 0.125)
        ref = sdpa_ref(qk[:, :C_].reshape(B, Lq, C_), qk[:, C_:].reshape(B, Lq, C_), v, H)
        assert_close(out.view(B, Lq, C_), ref, dtype, f"creeping maximum mode {mode}", k=6.0)
    finally:
        L.load().imh_debug_set(4, 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Lq", [(2, 20, 1024), (1, 33, 1024), (1, 43, 768), (2, 10, 4096)])
def test_attention_key_quarter_workgroups(L, dtype, B, H, Lq):
    """the pipelined kernel deals the items an XCD has beyond whole rounds of one per CU (per % 32 of them, up to 16: 8 of 40 at the
    L = 1024 layers of UNet batch 2, 16 of 80 at L = 4096; 1 of 33; 1 of 33 with a ragged last XCD) as workgroups of 32 queries whose
    four waves take the four quarters of the keys and merge in LDS: against the fp32 reference, bitwise repeatable, and within rounding
    of the whole-item launch (imh_debug_set(4, 7)); plus the creeping-maximum rows (a maximum that moves in every tile and every quarter)"""
    ctx = ctx_for(dtype)
    C_ = H * 64
    _attention_self_case(L, dtype, B, H, Lq)
    g = torch.Generator(device="cpu").manual_seed(7)
    q = torch.randn(B, Lq, H, 64, generator=g)
    k = torch.randn(B, Lq, H, 64, generator=g) * 0.2
    u = torch.nn.functional.normalize(torch.randn(64, generator=g), dim=0)
    q = q + 8.0 * u
    k = k + ((torch.arange(Lq) // 64).float() * (3.0 * 16 / (Lq // 64) / (0.125 * 8.0 * 1.4427)))[None, :, None, None] * u
    qk = torch.cat([q.reshape(B * Lq, C_), k.reshape(B * Lq, C_)], 1).to(dtype).to(DEV)
    v = rnd(B, Lq, C_, dtype=dtype, seed=2)
    vt = make_vt(v, Lq)
    ref = sdpa_ref(qk[:, :C_].reshape(B, Lq, C_), qk[:, C_:].reshape(B, Lq, C_), v, H)
    outs = []
    for mode in (0, 7):
        assert L.load().imh_debug_set(4, mode) == 0
        try:
            out = ctx.new(B * Lq, C_)
            ctx.attention(qk[:, :C_], qk[:, C_:], vt, out, B, H, Lq, Lq, Lq, 2 * C_, 2 * C_, B * Lq, C_, 0.125)
            assert_close(out.view(B, Lq, C_), ref, dtype, f"key-quarter workgroups, creeping maximum, mode {mode}", k=6.0)
            outs.append(out)
        finally:
            L.load().imh_debug_set(4, 0)
    assert not torch.equal(outs[0], outs[1]), "the two launches agree bit for bit: the key-quarter workgroups did not run"
    from conftest import rel_rms
    assert rel_rms(outs[0], outs[1]) < (6e-3 if dtype == torch.bfloat16 else 8e-4)


def _spiked_case(L):
    dtype = torch.bfloat16
    ctx = ctx_for(dtype)
    B, H, Lq = 1, 1, 256
    qk = rnd(B * Lq, 128, dtype=dtype, seed=1)
    qk[:, 64:][200] = qk[:, :64][7] * 6.0          # key 200 aligned with query 7 -> huge late score
    v = rnd(B, Lq, 64, dtype=dtype, seed=2)
    out = ctx.new(B * Lq, 64)
    ctx.attention(qk[:, :64], qk[:, 64:], make_vt(v, Lq), out, B, H, Lq, Lq, Lq, 128, 128, Lq, 64, 0.125)
    ref = sdpa_ref(qk[:, :64].reshape(B, Lq, 64), qk[:, 64:].reshape(B, Lq, 64), v, H)
    assert_close(out.view(B, Lq, 64), ref, dtype, "spiked", k=6.0)


# ------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,C_,silu", [(2, 256, 320, True), (1, 1024, 64, False), (2, 64, 1280, True), (1, 100, 2560, True),
                                          (2, 4096, 640, True)])
def test_groupnorm(L, dtype, B, HW, C_, silu):
    ctx = ctx_for(dtype)
    x = rnd(B, HW, C_, dtype=dtype, seed=1) + 0.5
    g, b = rnd(C_, dtype=dtype, seed=2) * 0.1 + 1, rnd(C_, dtype=dtype, seed=3) * 0.1
    y = ctx.groupnorm(x, g, b, 32, 1e-5, silu)
    ref = F.group_norm(x.float().transpose(1, 2), 32, g.float(), b.float(), 1e-5)
    if silu:
        ref = F.silu(ref)
    assert_close(y, ref.transpose(1, 2), dtype, "groupnorm")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C_", [(100, 640), (64, 1280), (7, 2048), (3, 4096), (5, 64)])
def test_layernorm(L, dtype, rows, C_):
    ctx = ctx_for(dtype)
    x = rnd(rows, C_, dtype=dtype, seed=1) * 2 + 0.3
    g, b = rnd(C_, dtype=dtype, seed=2) * 0.1 + 1, rnd(C_, dtype=dtype, seed=3) * 0.1
    y = ctx.layernorm(x, g, b, 1e-5)
    assert_close(y, F.layer_norm(x.float(), (C_,), g.float(), b.float(), 1e-5), dtype, "layernorm")


# ------------------------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise(L, dtype):
    ctx = ctx_for(dtype)
    # timestep embedding
    t = torch.tensor([958.0, 1.0, 500.0], device=DEV)
    y = ctx.new(3, 320)
    ctx.ew(L.EW_TIMESTEP, y, a=t, n=3, i=(320, 0, 0, 0, 0, 0))
    half = 160
    fr = torch.exp(-math.log(10000.0) * torch.arange(half, device=DEV, dtype=torch.float32) / half)
    a = t[:, None] * fr[None]
    assert_close(y, torch.cat([a.cos(), a.sin()], -1), dtype, "timestep", k=2.0)
    # silu + concat
    x = rnd(64, 128, dtype=dtype, seed=1)
    assert_close(ctx.silu(x), F.silu(x.float()), dtype, "silu", k=2.0)
    b = rnd(64, 64, dtype=dtype, seed=2)
    assert torch.equal(ctx.concat(x, b), torch.cat([x, b], -1))
    # conv_in with CFG duplication and input scale
    S, H, W, C0 = 2, 16, 12, 64
    lat = torch.randn(S, 4, H, W, device=DEV)
    w, bias = rnd(C0, 4, 3, 3, dtype=dtype, seed=3, scale=1 / 6), rnd(C0, dtype=dtype, seed=4)
    out = ctx.new(2 * S, H, W, C0)
    ctx.ew(L.EW_CONV_IN, out, a=lat, w=w, bias=bias, i=(S, H, W, C0, 2 * S, 0), f=(0.5, 0, 0, 0))
    xin = (lat * 0.5).to(dtype).float()
    ref = F.conv2d(torch.cat([xin, xin]), w.float(), bias.float(), padding=1).permute(0, 2, 3, 1)
    assert_close(out, ref, dtype, "conv_in")
    # CFG + scheduler step
    HW = H * W
    npred = rnd(2 * S, HW, 4, dtype=dtype, seed=5)
    lat2 = lat.clone()
    ctx.ew(L.EW_CFG_STEP, lat2, a=npred, i=(S, HW, 0, 1, 0, 0), f=(0.9, -0.3, 5.0, 0))
    n = npred.float().view(2, S, HW, 4).permute(0, 1, 3, 2)      # [2, S, 4, HW]
    eps = n[0] + 5.0 * (n[1] - n[0])
    ref = 0.9 * lat.view(S, 4, HW) + -0.3 * eps
    assert (lat2.view(S, 4, HW) - ref).abs().max().item() < 1e-5


def test_step_counter_tables(L):
    dtype = torch.bfloat16
    ctx = ctx_for(dtype)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    ttab = torch.tensor([958.0, 925.0, 892.0], device=DEV)
    y = ctx.new(2, 64)
    for i in range(3):
        ctx.ew(L.EW_TIMESTEP, y, a=ttab, step=step, n=2, i=(64, 0, 0, 0, 0, 0))
        assert abs(y[0, 0].float().item() - math.cos(ttab[i].item())) < 2 ** -7
        ctx.ew(L.EW_STEP_SET, step, i=(0, 0, 0, 0, 0, 0))
    assert step.item() == 3
    ctx.ew(L.EW_STEP_SET, step, i=(0, 1, 0, 0, 0, 0))
    assert step.item() == 0


# ------------------------------------------------------------------------------------ plans
def test_plan_record_replay_capture(L):
    from imagharmony_amd.ctx import Ctx
    dtype = torch.bfloat16
    x, w = rnd(256, 128, dtype=dtype, seed=1), rnd(192, 128, dtype=dtype, seed=2, scale=0.1)
    eager = ctx_for(dtype)
    ref = eager.layernorm(eager.gemm(x, w), None, None, 1e-5)
    rec = Ctx(DEV, dtype, record=True)
    y = rec.layernorm(rec.gemm(x, w), None, None, 1e-5)
    assert rec.lib.imh_plan_size(rec.plan) == 2
    rec.run()
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    y.zero_()
    rec.capture()
    rec.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    ms = rec.time_ops()
    assert len(ms) == 2 and all(m >= 0 for m in ms)


def test_errors_are_reported_not_thrown(L):
    ctx = ctx_for(torch.bfloat16)
    x, w = rnd(64, 96, dtype=torch.bfloat16, seed=1), rnd(64, 96, dtype=torch.bfloat16, seed=2)
    with pytest.raises(L.ImhError, match="multiple of 64"):
        ctx.gemm(x, w)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(64, 64), (128, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("shape", [(192, 256, 128), (300, 200, 1280), (2048, 1280, 640)])
def test_gemm_folded_layernorm(L, dtype, cfg, shape):
    """LN(x) W^T through the folded form -- gamma-scaled weights, row statistics from the row-statistics kernel (the caller
    passes none, Ctx supplies them) -- in both orientations (tokens as the X operand / as the W operand), against
    F.layer_norm + matmul in fp32, with a row mean of 2 sigma."""
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    M, N, K = shape
    x = (rnd(M, K, dtype=dtype, seed=1) * 1.5 + 3.0).contiguous()
    w = rnd(N, K, dtype=torch.float32, seed=2, scale=K ** -0.5)
    norm = torch.nn.LayerNorm(K, eps=1e-5)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3)))
        norm.bias.copy_(0.3 * torch.randn(K, generator=torch.Generator().manual_seed(4)))
    ref = F.layer_norm(x.float().cpu(), (K,), norm.weight, norm.bias, 1e-5) @ w.cpu().t()
    wg, s, c = fold_ln(w, norm, ctx)
    bm, bn = cfg
    y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5), cfg=(bm, bn, 1))
    assert_close(y, ref.to(DEV), dtype, f"folded LN (row form) {cfg}", k=6.0)
    if M % 16 == 0:                                          # the V^T layout permutes whole 16-key groups
        yt = ctx.gemm(wg, x, flags=L.GF_LN_COL | L.GF_VT_PERM, ln=(s, c, 1e-5), cfg=(bm, 128, 1))   # [N, M] = (LN(x) W^T)^T, V^T layout
        assert_close(vt_unpermute(yt), ref.t().to(DEV), dtype, f"folded LN (col form) {cfg}", k=6.0)
    else:
        yt = ctx.gemm(wg, x, flags=L.GF_LN_COL, ln=(s, c, 1e-5), cfg=(bm, 128, 1))                  # plain [N, M], ragged M
        assert_close(yt, ref.t().to(DEV), dtype, f"folded LN (col form, ragged) {cfg}", k=6.0)
    if M % 16 == 0:
        y2, yt2 = ctx.gemm_dual(dict(x=x, w=wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5)),
                                dict(x=wg, w=x, flags=L.GF_LN_COL | L.GF_VT_PERM, ln=(s, c, 1e-5)), cfg=(bm, 128))
        assert torch.equal(y2, ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5), cfg=(bm, 128, 1))) and torch.equal(yt2, yt)
    # GEGLU on top of the folded form (the ff.net.0 launch): interleaved (value, gate) columns
    if N % 32 == 0:
        g = ctx.gemm(x, wg, flags=L.GF_LN_ROW | L.GF_GEGLU, ln=(s, c, 1e-5), cfg=(bm, bn, 1))
        r = ref.to(DEV)
        assert_close(g, geglu_ref(r), dtype, f"folded LN + GEGLU {cfg}", k=8.0)
    with pytest.raises(L.ImhError, match="folded LayerNorm"):
        ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5), cfg=(64, 64, 2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,slots", [((256, 320, 64), 1), ((512, 640, 1280), 16), ((2048, 10240, 1280), 16), ((768, 960, 640), 8)])
def test_gemm_sixteen_wave_256x320_tile(L, dtype, shape, slots):
    """Round 6, csrc/gemm_w16.hip (variant 26256 x 320; tuning.json takes it for the ff.net.0 launches; its default form since the end of
    round 6 is eight fat waves, the sixteen-wave kernel stays as the A/B form): row-form folded LayerNorm with
    handed-over statistics, plain and + GEGLU, (a) against F.layer_norm + matmul (+ GEGLU) in fp32, (b) BIT-identical to the
    wave-specialised 256 x 160 kernel it replaces (same operands, same order of operations per output element), (c) bitwise repeatable;
    and the launch refuses what it does not implement (a bias, a residual, ragged tiles) with a status code."""
    from conftest import ref_row_stats
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    M, N, K = shape
    x = (rnd(M, K, dtype=dtype, seed=1) * 1.5 + 3.0).contiguous()
    w = rnd(N, K, dtype=torch.float32, seed=2, scale=K ** -0.5)
    norm = torch.nn.LayerNorm(K, eps=1e-5)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3)))
        norm.bias.copy_(0.3 * torch.randn(K, generator=torch.Generator().manual_seed(4)))
    ref = (F.layer_norm(x.float().cpu(), (K,), norm.weight, norm.bias, 1e-5) @ w.cpu().t()).to(DEV)
    wg, s, c = fold_ln(w, norm, ctx)
    st = (ref_row_stats(x.float(), slots).to(DEV), slots) if slots > 1 else ctx.row_stats(x)
    for flags, want, what in ((L.GF_LN_ROW, ref, "LN"), (L.GF_LN_ROW | L.GF_GEGLU, geglu_ref(ref), "LN + GEGLU")):
        y = ctx.gemm(x, wg, flags=flags, ln=(s, c, 1e-5, st), cfg=(26256, 320, 1))
        assert_close(y, want, dtype, f"sixteen-wave 256 x 320, {what} {shape}", k=8.0)
        assert torch.equal(y, ctx.gemm(x, wg, flags=flags, ln=(s, c, 1e-5, st), cfg=(23256, 160, 1))), f"{what}: differs from the 256 x 160 kernel"
        assert torch.equal(y, ctx.gemm(x, wg, flags=flags, ln=(s, c, 1e-5, st), cfg=(26256, 320, 1))), f"{what}: not repeatable"
        try:        # the default form is eight fat waves of 128 x 80 (gemm_f8_kernel); imh_debug_set(9, 0) = the sixteen 64 x 80 waves: the same bits
            ctx.lib.imh_debug_set(9, 0)
            y16 = ctx.gemm(x, wg, flags=flags, ln=(s, c, 1e-5, st), cfg=(26256, 320, 1))
        finally:
            ctx.lib.imh_debug_set(9, 1)
        assert torch.equal(y, y16), f"{what}: the eight-wave and the sixteen-wave forms differ"
    b = rnd(N, dtype=dtype, seed=5)
    with pytest.raises(L.ImhError, match="26256"):
        ctx.gemm(x, wg, bias=b, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), cfg=(26256, 320, 1))
    with pytest.raises(L.ImhError, match="26256"):
        ctx.gemm(x[:M - 64], wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, (st[0][:M - 64], st[1])), cfg=(26256, 320, 1))
    with pytest.raises(L.ImhError):
        ctx.gemm(x, wg, cfg=(26256, 320, 1))                      # no folded LayerNorm: not this variant's launch


def test_gemm_folded_layernorm_zero_variance_rows(L):
    """constant rows (variance 0): rstd = 1/sqrt(eps), finite output equal to the bias term W beta"""
    from imagharmony_amd.attention_processor import fold_ln
    dtype = torch.bfloat16
    ctx = ctx_for(dtype)
    M, N, K = 64, 128, 256
    x = torch.full((M, K), 2.0, dtype=dtype, device=DEV)
    w = rnd(N, K, dtype=torch.float32, seed=2, scale=K ** -0.5)
    norm = torch.nn.LayerNorm(K, eps=1e-5)
    with torch.no_grad():
        norm.bias.copy_(0.3 * torch.randn(K, generator=torch.Generator().manual_seed(4)))
    wg, s, c = fold_ln(w, norm, ctx)
    y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5))
    ref = F.layer_norm(x.float().cpu(), (K,), norm.weight, norm.bias, 1e-5) @ w.cpu().t()
    assert torch.isfinite(y.float()).all()
    assert (y.float().cpu() - ref).abs().max() < 0.05
