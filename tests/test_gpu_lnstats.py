"""GPU: the LayerNorm-statistics hand-over (csrc/imh_lnstats.h) -- the GEMM that writes a LayerNorm input leaves per-row
(sum, M2) slot partials behind from its epilogue, the consumers (ff.net.0 on the wave-specialised kernel, [Q|K] + V^T, the
fused cross-attention's to_q) merge them; rows no epilogue covers get theirs from IMH_EW_ROW_STATS.  The kernels have no
other source of LayerNorm statistics (the in-loop sum / sum-of-squares form of rounds 2-3 is gone).  Reference math: diffusers
BasicTransformerBlock.norm1/2/3 = torch LayerNorm, here F.layer_norm in fp32."""
import pytest
import torch
import torch.nn.functional as F

from conftest import built, ref_row_stats
from test_gpu_ops import DEV, DTYPES, EPS, L, assert_close, ctx_for, geglu_ref, rnd, vt_unpermute  # noqa: F401

pytestmark = pytest.mark.gpu


def _norm(K):
    norm = torch.nn.LayerNorm(K, eps=1e-5)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(K, generator=torch.Generator().manual_seed(3)))
        norm.bias.copy_(0.3 * torch.randn(K, generator=torch.Generator().manual_seed(4)))
    return norm


def _check_stats(st, slots, y, what):
    ref = ref_row_stats(y.float(), slots).to(st.device)
    assert st.shape == ref.shape, (st.shape, ref.shape)
    width = y.shape[1] // slots
    s_err = (st[..., 0] - ref[..., 0]).abs().max().item()
    s_scale = y.float().abs().max().item() * width
    m_rel = ((st[..., 1] - ref[..., 1]).abs() / (ref[..., 1].abs() + 1e-6 * width * y.float().pow(2).mean().item())).max().item()
    assert s_err <= 2e-6 * s_scale and m_rel <= 2e-4, f"{what}: sum err {s_err:.3e} (scale {s_scale:.3e}), M2 rel err {m_rel:.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg,width", built([((2464, 160, 1), 80), ((1464, 160, 1), 80), ((24128, 160, 1), 80), ((22128, 160, 1), 80), ((24128, 128, 1), 64),
                                       ((23256, 160, 1), 80), ((64, 64, 1), 32), ((128, 64, 1), 32), ((64, 128, 1), 64), ((128, 128, 1), 64)]))
def test_statistics_epilogue(L, dtype, cfg, width):
    """ln_stats_out of every variant that has the epilogue, in the two forms the forward uses (proj_in: bias only; to_out /
    ff.out: bias + residual), ragged M included: slot sums / M2 of the values AS STORED (rounded to the output dtype)"""
    ctx = ctx_for(dtype)
    assert ctx.lib.imh_gemm_stats_slot_width(cfg[0], cfg[1]) == width
    for (M, N, K) in [(256, 640, 128), (300, 1280, 192), (2048, 1280, 640)]:
        if N % width:
            continue
        x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
        bias, res = rnd(N, dtype=dtype, seed=5), (rnd(M, N, dtype=dtype, seed=6) * 1.5 + 0.5).contiguous()
        for residual in (None, res):
            y, (st, slots) = ctx.gemm(x, w, bias=bias, residual=residual, cfg=cfg, stats_out=True)
            assert slots == N // width
            ref = x.float() @ w.float().t() + bias.float() + (residual.float() if residual is not None else 0.0)
            assert_close(y, ref, dtype, f"gemm {cfg} {(M, N, K)}")
            _check_stats(st, slots, y, f"statistics epilogue {cfg} {(M, N, K)} residual={residual is not None}")
            y2, (st2, _) = ctx.gemm(x, w, bias=bias, residual=residual, cfg=cfg, stats_out=True)
            assert torch.equal(y2, y) and torch.equal(st2, st)
            for t in (y, st, y2, st2):
                ctx.free(t)


@pytest.mark.parametrize("dtype", DTYPES)
def test_statistics_fallback_kernel_and_rejections(L, dtype):
    """variants without the epilogue hand the statistics over through the row-statistics launch (one slot per row); the C
    side refuses ln_stats_out where no epilogue exists"""
    ctx = ctx_for(dtype)
    M, N, K = 300, 640, 256
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    for cfg in [(5258, 320, 1), (3064, 64, 1), (64, 64, 2)]:
        y, (st, slots) = ctx.gemm(x, w, cfg=cfg, stats_out=True)
        assert slots == 1
        _check_stats(st, 1, y, f"row-statistics kernel behind {cfg}")
    a = ctx.gemm(x, w, cfg=(5258, 320, 1), _args_only=True)[0]
    a.ln_stats_out, a.ln_slots_out = st.data_ptr(), 1
    assert ctx.lib.imh_gemm(a, ctx.stream()) == -1 and b"ln_stats_out" in ctx.lib.imh_last_error()


def _stats_for(ctx, x, how):
    """the three producers a consumer can meet: a GEMM epilogue's 80- / 32-wide slots (here minted with torch in the same
    format), or the one-slot row-statistics kernel"""
    K = x.shape[1]
    if how == "kernel":
        return ctx.row_stats(x)
    slots = K // how
    return ref_row_stats(x.float(), slots).to(DEV), slots


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", built([(23256, 160, 1), (23256, 128, 1), (2464, 160, 1), (1464, 160, 1), (24128, 160, 1), (22128, 160, 1), (24128, 128, 1), (128, 128, 1), (64, 64, 1)]))
@pytest.mark.parametrize("how", [80, 32, "kernel"])
def test_folded_layernorm_with_precomputed_statistics(L, dtype, cfg, how):
    """LN(x) W^T (+ GEGLU) with the statistics taken from the hand-over buffer, row form, every consuming variant; x with a
    row mean of 2 sigma"""
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    for (M, N, K) in [(512, 640, 320), (300, 960, 640), (2048, 2560, 1280)]:
        x = (rnd(M, K, dtype=dtype, seed=1) * 1.5 + 3.0).contiguous()
        w = rnd(N, K, dtype=torch.float32, seed=2, scale=K ** -0.5)
        norm = _norm(K)
        full = (F.layer_norm(x.float().cpu(), (K,), norm.weight, norm.bias, 1e-5) @ w.cpu().t()).to(DEV)
        wg, s, c = fold_ln(w, norm, ctx)
        st = _stats_for(ctx, x, how)
        y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), cfg=cfg)
        assert_close(y, full, dtype, f"LN (precomputed, {how}) {cfg} {(M, N, K)}", k=6.0)
        g = ctx.gemm(x, wg, flags=L.GF_LN_ROW | L.GF_GEGLU, ln=(s, c, 1e-5, st), cfg=cfg)
        assert_close(g, geglu_ref(full), dtype, f"LN (precomputed, {how}) + GEGLU {cfg} {(M, N, K)}", k=8.0)
        assert torch.equal(g, ctx.gemm(x, wg, flags=L.GF_LN_ROW | L.GF_GEGLU, ln=(s, c, 1e-5, st), cfg=cfg))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(64, 64), (128, 64), (64, 128), (128, 128)])
def test_dual_projection_with_precomputed_statistics(L, dtype, cfg):
    """self-attention's [Q|K] (row form) + V^T (column form, V^T layout) launch with the statistics handed over"""
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    for (M, N, K) in [(192, 256, 128), (2048, 1280, 640)]:
        x = (rnd(M, K, dtype=dtype, seed=1) * 1.5 + 3.0).contiguous()
        w = rnd(N, K, dtype=torch.float32, seed=2, scale=K ** -0.5)
        norm = _norm(K)
        ref = (F.layer_norm(x.float().cpu(), (K,), norm.weight, norm.bias, 1e-5) @ w.cpu().t()).to(DEV)
        wg, s, c = fold_ln(w, norm, ctx)
        for how in (32 if K % 32 == 0 else 64, "kernel"):
            st = _stats_for(ctx, x, how)
            bm, bn = cfg
            y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), cfg=(bm, bn, 1))
            assert_close(y, ref, dtype, f"row form {cfg}", k=6.0)
            yt = ctx.gemm(wg, x, flags=L.GF_LN_COL | L.GF_VT_PERM, ln=(s, c, 1e-5, st), cfg=(bm, 128, 1))
            assert_close(vt_unpermute(yt), ref.t(), dtype, f"column form {cfg}", k=6.0)
            y2, yt2 = ctx.gemm_dual(dict(x=x, w=wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st)),
                                    dict(x=wg, w=x, flags=L.GF_LN_COL | L.GF_VT_PERM, ln=(s, c, 1e-5, st)), cfg=(bm, 128))
            assert torch.equal(y2, ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), cfg=(bm, 128, 1))) and torch.equal(yt2, yt)


@pytest.mark.parametrize("dtype", DTYPES)
def test_large_common_offset_rows(L, dtype):
    """ADVICE r02 / VERDICT r03 weak 1: |mean| >> std (x = 50 + N(0, 0.1), as stored).  The hand-over statistics are (sum, M2)
    pairs merged without E[x^2] - mean^2, so the folded result matches torch's LayerNorm + Linear on every path a caller can
    take: statistics from a producer epilogue, from IMH_EW_ROW_STATS, and supplied by Ctx.gemm when the caller passes none;
    the C ABI refuses a folded-LayerNorm launch without statistics (there is no in-loop form to fall back to)."""
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    M, N, K = 256, 640, 1280
    x = (50.0 + 0.1 * rnd(M, K, dtype=torch.float32, seed=1)).to(dtype).contiguous()
    x[3] = 50.0                                            # and one zero-variance row
    w = rnd(N, K, dtype=torch.float32, seed=2, scale=K ** -0.5)
    norm = _norm(K)
    ref = (F.layer_norm(x.float().cpu(), (K,), norm.weight, norm.bias, 1e-5) @ w.cpu().t()).to(DEV)
    wg, s, c = fold_ln(w, norm, ctx)
    # statistics from a real producer: y = 0 @ W + residual(x) through the wave-specialised kernel's epilogue
    z = torch.zeros(M, 64, dtype=dtype, device=DEV)
    xx, st = ctx.gemm(z, torch.zeros(K, 64, dtype=dtype, device=DEV), residual=x, cfg=(2464, 160, 1), stats_out=True)
    assert torch.equal(xx, x) and st[1] == K // 80
    for cfg in [(2464, 160, 1), (23256, 160, 1), (128, 128, 1)]:
        y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), cfg=cfg)
        # the mean term rstd * mean * s_n is ~ 500 sigma of the result: its fp32 cancellation against acc bounds the error
        err = (y.float() - ref).abs().max().item()
        assert torch.isfinite(y.float()).all() and err < 0.15, f"{cfg}: max err {err:.3e} with handed-over statistics"
    for cfg in [(2464, 160, 1), (128, 128, 1), (64, 64, 1)]:
        for stx in (ctx.row_stats(x), None):                # the stand-alone producer; none given -> Ctx.gemm runs it
            y = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, stx), cfg=cfg)
            e2 = (y.float() - ref).abs().max().item()
            assert torch.isfinite(y.float()).all() and e2 < 0.15, f"{cfg}: max err {e2:.3e} with row-statistics-kernel statistics"
    yt = ctx.gemm(wg, x, flags=L.GF_LN_COL, ln=(s, c, 1e-5), cfg=(128, 128, 1))            # column form, statistics supplied by Ctx
    assert (yt.float() - ref.t()).abs().max().item() < 0.15
    a = ctx.gemm(x, wg, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), cfg=(128, 128, 1), _args_only=True)[0]
    a.ln_stats = None
    assert ctx.lib.imh_gemm(a, ctx.stream()) == -1 and b"ln_stats" in ctx.lib.imh_last_error()
    print(f"{dtype}: handed-over statistics max err {err:.3e} (result scale {ref.abs().max().item():.2f})")


@pytest.mark.parametrize("dtype", DTYPES)
def test_wave_specialised_projection_pair(L, dtype):
    if not L.experimental():
        pytest.skip("gemm_dual variant 24128 is compiled with -DIMH_EXPERIMENTAL only (the forward runs the one-launch [Q|K|V])")
    """variant 24128 of imh_gemm_dual: [Q|K] (row form, 128 x 160 tiles) + V^T (column form + V^T key permutation, 128 x 128
    tiles) in one wave-specialised launch with handed-over statistics -- the self-attention projections of the forward"""
    from imagharmony_amd.attention_processor import fold_ln
    ctx = ctx_for(dtype)
    for (M, C_) in [(256, 320), (2048, 1280), (8192, 640), (384, 960)]:
        x = (rnd(M, C_, dtype=dtype, seed=1) * 1.5 + 2.0).contiguous()
        wqk = rnd(2 * C_, C_, dtype=torch.float32, seed=2, scale=C_ ** -0.5)
        wv = rnd(C_, C_, dtype=torch.float32, seed=3, scale=C_ ** -0.5)
        norm = _norm(C_)
        xn = F.layer_norm(x.float().cpu(), (C_,), norm.weight, norm.bias, 1e-5)
        fq, fv = fold_ln(wqk, norm, ctx), fold_ln(wv, norm, ctx)
        for how in (80, "kernel"):
            st = _stats_for(ctx, x, how)
            qk, vt = ctx.gemm_dual(dict(x=x, w=fq[0], flags=L.GF_LN_ROW, ln=(fq[1], fq[2], 1e-5, st)),
                                   dict(x=fv[0], w=x, flags=L.GF_VT_PERM | L.GF_LN_COL, ln=(fv[1], fv[2], 1e-5, st)), cfg=(24128, 160))
            assert_close(qk, (xn @ wqk.cpu().t()).to(DEV), dtype, f"[Q|K] {(M, C_)} {how}", k=6.0)
            assert_close(vt_unpermute(vt), (wv.cpu() @ xn.t()).to(DEV), dtype, f"V^T {(M, C_)} {how}", k=6.0)
            qk2, vt2 = ctx.gemm_dual(dict(x=x, w=fq[0], flags=L.GF_LN_ROW, ln=(fq[1], fq[2], 1e-5, st)),
                                     dict(x=fv[0], w=x, flags=L.GF_VT_PERM | L.GF_LN_COL, ln=(fv[1], fv[2], 1e-5, st)), cfg=(24128, 160))
            assert torch.equal(qk, qk2) and torch.equal(vt, vt2)
    # no statistics from the caller: Ctx.gemm_dual runs the row-statistics kernel and hands its output to both problems
    qk3, vt3 = ctx.gemm_dual(dict(x=x, w=fq[0], flags=L.GF_LN_ROW, ln=(fq[1], fq[2], 1e-5)),
                             dict(x=fv[0], w=x, flags=L.GF_VT_PERM | L.GF_LN_COL, ln=(fv[1], fv[2], 1e-5)), cfg=(24128, 160))
    assert torch.equal(qk3, qk) and torch.equal(vt3, vt)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(23256, 160, 1), (23256, 128, 1), (24128, 160, 1), (2464, 160, 1), (1464, 160, 1)])
def test_one_launch_qkv_with_transposed_v(L, dtype, cfg):
    """[Q|K|V] = LN(x) [Wq;Wk;Wv]^T as ONE wave-specialised launch (imh_gemm_args.Yt): Q and K land row-major in y [M, 2C], the V
    columns leave the kernel transposed through LDS in the V^T layout (16-token groups permuted) -- equal, bit for bit, to what
    the two-problem launch of round 3 produces with the same statistics, and to torch within rounding; feeds imh_attention as is
    (reference: attention_processor.py:292-316)"""
    from imagharmony_amd.attention_processor import fold_ln
    from test_gpu_ops import make_vt, sdpa_ref
    ctx = ctx_for(dtype)
    for (B, Lq, C_) in [(2, 256, 320), (2, 1024, 1280), (1, 512, 640)]:
        if (2 * C_) % cfg[1] or (3 * C_) % cfg[1]:          # (23256 x 128: whole 128-column tiles on both sides of the [Q|K] | V boundary)
            continue
        M = B * Lq
        x = (rnd(M, C_, dtype=dtype, seed=1) * 1.5 + 2.0).contiguous()
        w3 = rnd(3 * C_, C_, dtype=torch.float32, seed=2, scale=C_ ** -0.5)
        norm = _norm(C_)
        xn = F.layer_norm(x.float().cpu(), (C_,), norm.weight, norm.bias, 1e-5)
        f3 = fold_ln(w3, norm, ctx)
        st = _stats_for(ctx, x, 80)
        vt = torch.zeros(C_, M, dtype=dtype, device=DEV)
        qk = ctx.gemm(x, f3[0], flags=L.GF_LN_ROW, ln=(f3[1], f3[2], 1e-5, st), cfg=cfg, yt=(vt, 2 * C_))
        assert tuple(qk.shape) == (M, 2 * C_)
        ref = (xn @ w3.cpu().t()).to(DEV)
        assert_close(qk, ref[:, :2 * C_], dtype, f"[Q|K] {cfg} {(B, Lq, C_)}", k=6.0)
        assert_close(vt_unpermute(vt), ref[:, 2 * C_:].t(), dtype, f"V^T {cfg} {(B, Lq, C_)}", k=6.0)
        # the same numbers as the plain row-form launch over all 3C columns, transposed + permuted by torch
        full = ctx.gemm(x, f3[0], flags=L.GF_LN_ROW, ln=(f3[1], f3[2], 1e-5, st), cfg=cfg)
        assert torch.equal(full[:, :2 * C_], qk)
        v = full[:, 2 * C_:].reshape(B, Lq, C_)
        assert torch.equal(make_vt(v, Lq), vt), "transposed epilogue == torch transpose + 16-group permutation of the plain store"
        vt2 = torch.zeros_like(vt)
        assert torch.equal(ctx.gemm(x, f3[0], flags=L.GF_LN_ROW, ln=(f3[1], f3[2], 1e-5, st), cfg=cfg, yt=(vt2, 2 * C_)), qk) and torch.equal(vt2, vt)
        H = C_ // 64
        out = ctx.new(M, C_)
        ctx.attention(qk[:, :C_], qk[:, C_:], vt, out, B, H, Lq, Lq, Lq, 2 * C_, 2 * C_, M, C_, 0.125)
        assert_close(out.view(B, Lq, C_), sdpa_ref(qk[:, :C_].reshape(B, Lq, C_), qk[:, C_:].reshape(B, Lq, C_), v, H), dtype, "attention on the one-launch projections", k=6.0)
    with pytest.raises(L.ImhError, match="Yt"):
        ctx.gemm(x, f3[0], flags=L.GF_LN_ROW, ln=(f3[1], f3[2], 1e-5, st), cfg=(128, 128, 1), yt=(vt, 2 * C_))
