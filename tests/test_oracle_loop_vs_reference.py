"""CPU, build container only: the reference's OWN denoise loop (ip_adapter/custom_pipelines.py __call__, executed
verbatim through oracle/refshim.py on a restated base class) against the oracle's restatement of it
(oracle/pipeline.py).  Pins row a11 of SURVEY.md section 8 -- scale gating window, CFG order [uncond | cond], guidance
rescale, micro-conditioning time_ids, scheduler call order -- to the reference code itself.  The UNet and the
scheduler are the oracle's (diffusers is absent), installed with the REFERENCE's processor classes."""
import pytest
import torch

from oracle import refshim
from oracle.detfill import det_fill, det_randn
from oracle.pipeline import denoise as oracle_denoise
from oracle.schedulers import DDIMScheduler, EulerDiscreteScheduler
from oracle.sdxl_unet import UNet2DConditionModel, tiny_config

pytestmark = pytest.mark.skipif(not refshim.available(), reason="reference tree not present")


class _Cfg:
    num_train_timesteps = 1000


def _sched(kind):
    s = DDIMScheduler() if kind == "ddim" else EulerDiscreteScheduler()
    orig = s.set_timesteps
    s.set_timesteps = lambda n, device=None: orig(n)            # the reference passes device=
    s.order = 1
    s.config = _Cfg()
    orig_step = s.step
    s.step = lambda eps, t, lat, return_dict=False, **kw: orig_step(eps, t, lat)
    return s


def _unet_with_reference_processors(ap):
    cfg = tiny_config()
    u = det_fill(UNet2DConditionModel(cfg), 5).eval()
    procs = {}
    for name in u.attn_processors.keys():                       # ip_adapter.py:102-123
        if name.endswith("attn1.processor"):
            procs[name] = ap.AttnProcessor2_0()
        else:
            hidden = {"mid_block": cfg.block_out_channels[-1]}.get(name.split(".")[0])
            if hidden is None:
                bid = int(name.split(".")[1])
                hidden = (list(reversed(cfg.block_out_channels)) if name.startswith("up_blocks") else cfg.block_out_channels)[bid]
            skip = "down_blocks.2.attentions.1" not in name
            p = ap.IPAttnProcessor2_0(hidden_size=hidden, cross_attention_dim=cfg.cross_attention_dim, scale=0.8,
                                      num_tokens=4, skip=skip)
            procs[name] = det_fill(p, 7, prefix=name)
    u.set_attn_processor(procs)
    return u, cfg


@pytest.mark.parametrize("kind,kw", [("ddim", {}), ("euler", {}),
                                     ("ddim", dict(control_guidance_start=0.3, control_guidance_end=0.7)),
                                     ("ddim", dict(denoising_end=0.6)),
                                     ("ddim", dict(guidance_rescale=0.7, original_size=(512, 384), crops_coords_top_left=(16, 32),
                                                   target_size=(256, 256)))])
def test_reference_loop_equals_oracle_loop(kind, kw):
    Pipe, ap = refshim.load_pipeline_class()
    u, cfg = _unet_with_reference_processors(ap)

    class _U(torch.nn.Module):                                   # the reference passes cross_attention_kwargs / return_dict
        def __init__(s):
            super().__init__()
            s.u = u
            s.config = type("C", (), {"in_channels": cfg.in_channels})()

        @property
        def attn_processors(s):
            return s.u.attn_processors

        def forward(s, x, t, encoder_hidden_states=None, cross_attention_kwargs=None, added_cond_kwargs=None, return_dict=False):
            return s.u(x, t, encoder_hidden_states=encoder_hidden_states, added_cond_kwargs=added_cond_kwargs)

    pipe = Pipe.__new__(Pipe)
    pipe.unet, pipe.scheduler, pipe.default_sample_size, pipe.watermark = _U(), _sched(kind), 32, None
    lat = det_randn((1, 4, 32, 32), 3)
    cd = cfg.cross_attention_dim
    pe, ne = det_randn((1, 81, cd), 4), det_randn((1, 81, cd), 5)
    po, no = det_randn((1, cfg.pooled_dim), 6), det_randn((1, cfg.pooled_dim), 7)
    with torch.no_grad():
        ref = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=po, negative_pooled_prompt_embeds=no,
                   height=256, width=256, num_inference_steps=4, guidance_scale=5.0, latents=lat.clone(),
                   output_type="latent", **kw).images
        # the oracle loop on the same UNet object (the oracle's set_scale goes through its own isinstance checks, so
        # hand it a UNet with the ORACLE processor classes carrying the same weights)
        from oracle import modules as om
        from oracle.pipeline import install_ip_processors
        u2 = det_fill(UNet2DConditionModel(cfg), 5).eval()
        for n, p in install_ip_processors(u2, num_tokens=4, scale=0.8).items():
            if isinstance(p, om.IPAttnProcessor2_0):
                det_fill(p, 7, prefix=n)
        mine = oracle_denoise(u2, DDIMScheduler() if kind == "ddim" else EulerDiscreteScheduler(), lat.clone(), pe, ne, po, no,
                              256, 256, num_inference_steps=4, guidance_scale=5.0, **kw)
    assert ref.shape == mine.shape
    err = ((ref - mine).pow(2).mean().sqrt() / mine.pow(2).mean().sqrt()).item()
    assert err < 1e-5, err
