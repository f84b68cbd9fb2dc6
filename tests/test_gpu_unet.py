"""GPU parity of the fused UNet forward (imagharmony_amd.unet) against the CPU fp32 oracle
(oracle.sdxl_unet + oracle.modules processors) on the reduced-width config: identical seeded
weights, identical inputs.  Tolerance: rel-RMS of the noise prediction <= 3e-2 in bf16 and
<= 6e-3 in fp16 (per-module noise 4.8e-3 / 6e-4, SURVEY.md 4, compounded over ~40 layers)."""
import pytest
import torch

from conftest import rel_rms
from oracle import modules as om
from oracle.detfill import det_fill, det_randn
from oracle.pipeline import install_ip_processors
from oracle.sdxl_unet import UNet2DConditionModel as OracleUNet
from oracle.sdxl_unet import tiny_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float16: 6e-3, torch.bfloat16: 3e-2}


def build_pair(dtype, num_tokens=4, scale=0.8):
    from imagharmony_amd.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
    from imagharmony_amd.unet import UNet2DConditionModel, UNetConfig
    ocfg = tiny_config()
    with torch.no_grad():
        ou = det_fill(OracleUNet(ocfg), 5).eval()
        procs = install_ip_processors(ou, num_tokens=num_tokens, scale=scale)
        for n, p in procs.items():
            if isinstance(p, om.IPAttnProcessor2_0):
                det_fill(p, 7, prefix=n)
    cfg = UNetConfig(**{k: getattr(ocfg, k) for k in UNetConfig.__dataclass_fields__})
    hu = UNet2DConditionModel(cfg)
    hp = {}
    for name, p in procs.items():
        hp[name] = AttnProcessor2_0() if isinstance(p, om.AttnProcessor2_0) else \
            IPAttnProcessor2_0(p.hidden_size, p.cross_attention_dim, scale=p.scale, num_tokens=p.num_tokens, skip=p.skip)
    hu.set_attn_processor(hp)
    missing, unexpected = hu.load_state_dict(ou.state_dict(), strict=True)
    hu = hu.to(DEV, dtype)
    return ou, hu, ocfg


def inputs(ocfg, B=2, T=4, hw=32):
    """BASELINE.json shape families on the reduced-width UNet: T image tokens appended after 77 text tokens"""
    x = det_randn((B, 4, hw, hw), 3)
    ehs = det_randn((B, 77 + T, ocfg.cross_attention_dim), 4)
    te = det_randn((B, ocfg.pooled_dim), 6)
    ids = torch.tensor([[hw * 8, hw * 8, 0, 0, hw * 8, hw * 8]], dtype=torch.float32).repeat(B, 1)
    return x, ehs, te, ids


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_unet_forward_matches_oracle(dtype):
    ou, hu, ocfg = build_pair(dtype)
    x, ehs, te, ids = inputs(ocfg)
    with torch.no_grad():
        ref = ou(x, torch.tensor(500.0), ehs, added_cond_kwargs={"text_embeds": te, "time_ids": ids})[0]
    y = hu(x.to(DEV), torch.tensor(500.0), ehs.to(DEV, dtype),
           added_cond_kwargs={"text_embeds": te.to(DEV, dtype), "time_ids": ids.to(DEV)})[0]
    assert y.shape == ref.shape
    r = rel_rms(y.float().cpu(), ref)
    print(f"unet tiny {dtype}: rel-rms {r:.3e}")
    assert r < TOL[dtype], f"rel-rms {r:.3e}"


def test_unet_recorded_plan_equals_eager():
    """the recorded plan / hipGraph replay computes exactly what the eager per-op path does"""
    from imagharmony_amd.ctx import Ctx
    dtype = torch.bfloat16
    ou, hu, ocfg = build_pair(dtype)
    x, ehs, te, ids = inputs(ocfg)
    y_eager = hu(x.to(DEV), torch.tensor(321.0), ehs.to(DEV, dtype),
                 added_cond_kwargs={"text_embeds": te.to(DEV, dtype), "time_ids": ids.to(DEV)})[0]
    pre = Ctx(DEV, dtype)
    st = hu.prepare_conditioning(pre, ehs.to(DEV, dtype), te.to(DEV, dtype), ids.to(DEV))
    st.t_value = torch.full((2,), 321.0, device=DEV)
    st.latents = x.to(DEV).float().contiguous()
    rec = Ctx(DEV, dtype, record=True)
    out = hu.emit_forward(rec, st, 2, 32, 32, cfg_dup=False)
    rec.run()
    torch.cuda.synchronize()
    y_plan = out.view(2, 32, 32, 4).permute(0, 3, 1, 2).clone()
    assert torch.equal(y_plan, y_eager)
    out.zero_()
    rec.capture()
    rec.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(2, 32, 32, 4).permute(0, 3, 1, 2), y_eager)


@pytest.mark.parametrize("dtype,B,T", [(torch.float16, 4, 16), (torch.bfloat16, 2, 32), (torch.bfloat16, 1, 4)])
def test_unet_forward_batch_and_ip_token_variants(dtype, B, T):
    """configs[3]-like (batch 4, Resampler num_queries=16, fp16) and configs[4]-like (two 16-token embeds = 32
    image tokens) on the reduced-width UNet, plus an odd batch."""
    ou, hu, ocfg = build_pair(dtype, num_tokens=T)
    x, ehs, te, ids = inputs(ocfg, B=B, T=T)
    with torch.no_grad():
        ref = ou(x, torch.tensor(77.0), ehs, added_cond_kwargs={"text_embeds": te, "time_ids": ids})[0]
    y = hu(x.to(DEV), torch.tensor(77.0), ehs.to(DEV, dtype),
           added_cond_kwargs={"text_embeds": te.to(DEV, dtype), "time_ids": ids.to(DEV)})[0]
    r = rel_rms(y.float().cpu(), ref)
    assert r < TOL[dtype], f"B={B} T={T}: rel-rms {r:.3e}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_unet_forward_with_standalone_layernorm(dtype, monkeypatch):
    """A/B of the default path (LayerNorm folded into the consumer GEMMs, statistics inside their K loops) against
    the stand-alone LayerNorm kernels: both meet the oracle tolerance and agree with each other to rounding"""
    import imagharmony_amd.unet as hunet
    ou, hu, ocfg = build_pair(dtype)
    x, ehs, te, ids = inputs(ocfg)
    with torch.no_grad():
        ref = ou(x, torch.tensor(500.0), ehs, added_cond_kwargs={"text_embeds": te, "time_ids": ids})[0]
    run = lambda: hu(x.to(DEV), torch.tensor(500.0), ehs.to(DEV, dtype),
                     added_cond_kwargs={"text_embeds": te.to(DEV, dtype), "time_ids": ids.to(DEV)})[0].float().cpu()
    assert hunet.FOLD_LAYERNORM
    y_fold = run()
    monkeypatch.setattr(hunet, "FOLD_LAYERNORM", False)
    y_ln = run()
    r1, r2, r12 = rel_rms(y_fold, ref), rel_rms(y_ln, ref), rel_rms(y_fold, y_ln)
    print(f"unet tiny {dtype}: folded-LN {r1:.3e}, stand-alone LN {r2:.3e}, between them {r12:.3e}")
    assert r1 < TOL[dtype] and r2 < TOL[dtype] and r12 < TOL[dtype]
