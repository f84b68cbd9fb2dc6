"""CPU: structural pins of the restated SDXL UNet (diffusers is absent) and
closed-form pins of the restated schedulers."""
import numpy as np
import torch

from oracle.pipeline import denoise, install_ip_processors
from oracle.schedulers import DDIMScheduler, EulerDiscreteScheduler
from oracle.sdxl_unet import UNet2DConditionModel, sdxl_config, tiny_config
from oracle.detfill import det_fill, det_randn
from oracle import modules as om


def test_sdxl_param_count_and_processor_schema():
    with torch.device("meta"):
        m = UNet2DConditionModel(sdxl_config())
    assert sum(p.numel() for p in m.parameters()) == 2_567_463_684
    ap = m.attn_processors
    assert len(ap) == 140
    assert sum(k.endswith("attn1.processor") for k in ap) == 70
    active = [k for k in ap if "down_blocks.2.attentions.1" in k and k.endswith("attn2.processor")]
    assert len(active) == 10
    keys = list(ap)
    # registration order down_blocks -> up_blocks -> mid_block; attn2 sits at the odd positions
    assert keys[0].startswith("down_blocks.1.") and keys[-1].startswith("mid_block.")
    assert all(keys[i].endswith("attn2.processor") for i in range(1, 140, 2))
    sd = m.state_dict()
    for k in ["conv_in.weight", "time_embedding.linear_1.weight", "add_embedding.linear_2.bias",
              "down_blocks.0.resnets.1.time_emb_proj.weight", "down_blocks.0.downsamplers.0.conv.weight",
              "down_blocks.1.resnets.0.conv_shortcut.weight", "down_blocks.2.attentions.1.proj_in.weight",
              "down_blocks.2.attentions.1.transformer_blocks.9.attn2.to_out.0.bias",
              "mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
              "mid_block.resnets.1.norm2.weight", "up_blocks.0.upsamplers.0.conv.bias",
              "up_blocks.1.attentions.2.transformer_blocks.1.ff.net.2.weight",
              "up_blocks.2.resnets.2.conv_shortcut.bias", "conv_norm_out.weight", "conv_out.bias"]:
        assert k in sd, k
    assert sd["up_blocks.0.resnets.2.conv1.weight"].shape == (1280, 1920, 3, 3)
    assert sd["up_blocks.2.resnets.0.conv1.weight"].shape == (320, 960, 3, 3)
    assert sd["add_embedding.linear_1.weight"].shape == (1280, 2816)


def test_sdxl_unet_state_dict_equals_the_published_manifest():
    """set-equality of (key, shape) -- not counts -- between the restated diffusers UNet (oracle), the product's
    UNet2DConditionModel and tests/golden/sdxl_unet_manifest.txt, the state-dict manifest of stabilityai/stable-diffusion-xl-base-1.0's
    `unet` enumerated from its published config.json by string templates (oracle/gen_manifest.py: a different mechanism from either
    module tree; it reproduces the published 2,567,463,684 parameters).  A renamed, missing, extra or mis-shaped tensor in either model
    fails here -- a real checkpoint loads into both with strict=True."""
    import os
    from conftest import GOLDEN
    from oracle.gen_manifest import numel, read, unet_manifest
    man = read(os.path.join(GOLDEN, "sdxl_unet_manifest.txt"))
    assert man == dict(unet_manifest()) and len(man) == 1680                       # the committed fixture is what the generator writes
    assert numel(man.items()) == 2_567_463_684
    with torch.device("meta"):
        o = UNet2DConditionModel(sdxl_config())
    osd = {k: tuple(v.shape) for k, v in o.state_dict().items()}
    assert set(osd) == set(man), (sorted(set(osd) - set(man))[:5], sorted(set(man) - set(osd))[:5])
    assert osd == man
    from imagharmony_amd.unet import UNet2DConditionModel as HU, UNetConfig
    with torch.device("meta"):
        h = HU(UNetConfig())
    hsd = {k: tuple(v.shape) for k, v in h.state_dict().items()}
    assert set(hsd) == set(man), (sorted(set(hsd) - set(man))[:5], sorted(set(man) - set(hsd))[:5])
    assert hsd == man


def test_added_conditioning_shapes_follow_the_clip_text_encoders():
    """add_embedding.linear_1 takes [pooled text embedding | 6 size / crop ids x 256 sinusoids] (custom_pipelines.py:283-301 ->
    diffusers _get_add_time_ids): the pooled embedding is text_encoder_2's `text_embeds` (OpenCLIP bigG: projection_dim 1280), the
    cross-attention context the concat of the two encoders' penultimate hidden states (768 + 1280 = 2048).  Checked against stock
    `transformers` CLIP modules built from those published config values (shape facts only: no weights offline)."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    c1 = CLIPTextConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, projection_dim=768)
    c2 = CLIPTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20, projection_dim=1280, hidden_act="gelu")
    with torch.device("meta"):
        te1, te2 = CLIPTextModel(c1), CLIPTextModelWithProjection(c2)
    cfg = sdxl_config()
    assert te2.text_projection.out_features == cfg.pooled_dim == 1280
    assert cfg.projection_class_embeddings_input_dim == te2.text_projection.out_features + 6 * cfg.addition_time_embed_dim == 2816
    assert te1.config.hidden_size + te2.config.hidden_size == cfg.cross_attention_dim == 2048
    from imagharmony_amd.unet import UNetConfig
    h = UNetConfig()
    assert (h.pooled_dim, h.cross_attention_dim, h.projection_class_embeddings_input_dim) == (1280, 2048, 2816)


def test_ip_adapter_state_dict_keys_like_convert_bin():
    """ModuleList(unet.attn_processors.values()) keys are '<odd idx>.to_k_ip.weight'
    (ip_adapter/ip_adapter.py:153-154, convert_bin.py:21-40)."""
    with torch.device("meta"):
        m = UNet2DConditionModel(tiny_config())
        install_ip_processors(m, num_tokens=4)
        ml = torch.nn.ModuleList(m.attn_processors.values())
    ks = list(ml.state_dict().keys())
    n2 = sum(k.endswith("attn2.processor") for k in m.attn_processors)
    assert len(ks) == 2 * n2
    assert ks[0] == "1.to_k_ip.weight" and ks[1] == "1.to_v_ip.weight"


def test_ddim_tables_and_identities():
    s = DDIMScheduler()
    s.set_timesteps(30)
    ts = s.timesteps.tolist()
    assert ts[0] == 958 and ts[1] == 925 and ts[-1] == 1 and len(ts) == 30
    ac = s.alphas_cumprod
    assert abs(float(ac[0]) - (1 - 0.00085)) < 1e-6 and abs(float(ac[-1]) - 0.0046596) < 2e-5
    # exactness: if eps is the true noise, one step lands on sqrt(a')x0 + sqrt(1-a')eps
    x0, eps = det_randn((1, 4, 8, 8), 1), det_randn((1, 4, 8, 8), 2)
    t = 958
    a = ac[t]
    xt = a.sqrt() * x0 + (1 - a).sqrt() * eps
    ap = ac[t - 33]
    assert torch.allclose(s.step(eps, t, xt)[0], ap.sqrt() * x0 + (1 - ap).sqrt() * eps, atol=2e-5)


def test_ddim_and_euler_closed_forms_at_every_timestep():
    """the exact-recovery identities at ALL timesteps of the 30- and 50-step schedules (configs[1] / configs[3]),
    not at one.  DDIM (eta = 0): with the true noise as the prediction, x_t = sqrt(a_t) x0 + sqrt(1 - a_t) eps steps to
    sqrt(a_prev) x0 + sqrt(1 - a_prev) eps.  Euler: x = x0 + sigma eps steps to x0 + sigma_next eps, the model input is x / sqrt(sigma^2
    + 1), and the sigmas are sqrt((1 - a) / a) at the (leading-spaced, offset 1) timesteps."""
    x0, eps = det_randn((1, 4, 8, 8), 1).double(), det_randn((1, 4, 8, 8), 2).double()
    for n in (30, 50):
        s = DDIMScheduler()
        s.set_timesteps(n)
        r = 1000 // n
        ts = s.timesteps.tolist()
        assert ts == [i * r + 1 for i in range(n - 1, -1, -1)]
        ac = s.alphas_cumprod.double()
        for t in ts:
            a = ac[t]
            ap = ac[t - r] if t - r >= 0 else ac[0]
            xt = a.sqrt() * x0 + (1 - a).sqrt() * eps
            got = s.step(eps.float(), t, xt.float())[0].double()
            assert torch.allclose(got, ap.sqrt() * x0 + (1 - ap).sqrt() * eps, atol=3e-5), (n, t)
        e = EulerDiscreteScheduler()
        e.set_timesteps(n)
        sig = e.sigmas.double()
        assert [int(t) for t in e.timesteps.tolist()] == ts
        want = ((1 - ac) / ac).sqrt()
        for i, t in enumerate(e.timesteps):
            assert abs(float(sig[i]) - float(want[int(t)])) < 1e-5 * max(1.0, float(sig[i]))
            x = x0 + sig[i] * eps
            assert torch.allclose(e.scale_model_input(x.float(), t).double(), x / (sig[i] ** 2 + 1).sqrt(), atol=1e-5)
            got = e.step(eps.float(), t, x.float())[0].double()
            assert torch.allclose(got, x0 + sig[i + 1] * eps, atol=2e-4 * max(1.0, float(sig[i]))), (n, i)


def test_euler_tables():
    s = EulerDiscreteScheduler()
    s.set_timesteps(30)
    assert s.timesteps[0] == 958 and len(s.sigmas) == 31 and float(s.sigmas[-1]) == 0.0
    assert abs(s.init_noise_sigma - float((s.sigmas[0] ** 2 + 1) ** 0.5)) < 1e-6
    assert np.all(np.diff(s.sigmas.numpy()) < 0)


def test_tiny_unet_denoise_runs_and_scale_gating():
    torch.manual_seed(0)
    cfg = tiny_config()
    with torch.no_grad():
        unet = det_fill(UNet2DConditionModel(cfg), 5).eval()
        procs = install_ip_processors(unet, num_tokens=4, scale=1.0)
        for n, p in procs.items():
            if isinstance(p, om.IPAttnProcessor2_0):
                det_fill(p, 7, prefix=n)
        lat = det_randn((1, 4, 16, 16), 3)
        pe, ne = det_randn((1, 81, cfg.cross_attention_dim), 4), det_randn((1, 81, cfg.cross_attention_dim), 5)
        po, no = det_randn((1, cfg.pooled_dim), 6), det_randn((1, cfg.pooled_dim), 7)
        a = denoise(unet, DDIMScheduler(), lat, pe, ne, po, no, 128, 128, num_inference_steps=3)
        b = denoise(unet, DDIMScheduler(), lat, pe, ne, po, no, 128, 128, num_inference_steps=3,
                    control_guidance_end=0.0)       # IP scale gated off on every step
        assert a.shape == lat.shape and torch.isfinite(a).all()
        assert (a - b).abs().max() > 1e-4           # the IP branch matters


def test_oracle_unet_matches_its_committed_fixture():
    """tests/golden/oracle_tiny_unet.pt was produced by the oracle itself (oracle/gen_golden.py: diffusers cannot be
    executed here, so this guards the restatement against drift; it is not a reference pin)."""
    import os
    from conftest import GOLDEN
    from oracle.gen_golden import unet_fixture
    g = torch.load(os.path.join(GOLDEN, "oracle_tiny_unet.pt"))
    with torch.no_grad():
        now = unet_fixture()
    for k in g:
        a, b = now[k].float(), g[k].float()
        assert ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()) < 2e-3, k


def test_rescale_noise_cfg_closed_form():
    """restated diffusers rescale_noise_cfg (arXiv 2305.08891 3.4): phi = 1 gives the CFG prediction the per-sample
    std of the text-conditioned one; phi = 0 is the identity; samples are independent"""
    from oracle.pipeline import rescale_noise_cfg
    g = torch.Generator().manual_seed(0)
    text = torch.randn(3, 4, 8, 8, generator=g) * torch.tensor([1.0, 2.0, 0.5]).view(3, 1, 1, 1)
    cfg = torch.randn(3, 4, 8, 8, generator=g) * 3.0 + 0.2
    full = rescale_noise_cfg(cfg, text, 1.0)
    assert torch.allclose(full.flatten(1).std(1), text.flatten(1).std(1), rtol=1e-5)
    assert torch.equal(rescale_noise_cfg(cfg, text, 0.0), cfg)
    half = rescale_noise_cfg(cfg, text, 0.5)
    assert torch.allclose(half, 0.5 * full + 0.5 * cfg, atol=1e-6)
    assert torch.allclose(rescale_noise_cfg(cfg[1:2], text[1:2], 0.7), rescale_noise_cfg(cfg, text, 0.7)[1:2], atol=1e-6)
