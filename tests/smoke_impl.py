"""smoke(): one tiny invocation of the hot path on the GPU (reduced-width SDXL UNet, 2 DDIM steps with
CFG and the IP branch) checked against the CPU oracle.  Imported by __graft_entry__.smoke() only."""
import torch

from oracle import modules as om
from oracle.detfill import det_fill, det_randn
from oracle.pipeline import denoise as oracle_denoise
from oracle.pipeline import install_ip_processors as oracle_install
from oracle.schedulers import DDIMScheduler as OracleDDIM
from oracle.sdxl_unet import UNet2DConditionModel as OracleUNet
from oracle.sdxl_unet import tiny_config


def build_pair(device, dtype, num_tokens=4, scale=0.8):
    from imagharmony_amd.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
    from imagharmony_amd.unet import UNet2DConditionModel, UNetConfig
    ocfg = tiny_config()
    with torch.no_grad():
        ou = det_fill(OracleUNet(ocfg), 5).eval()
        procs = oracle_install(ou, num_tokens=num_tokens, scale=scale)
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
    hu.load_state_dict(ou.state_dict(), strict=True)
    return ou, hu.to(device, dtype), ocfg


def rel_rms(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def denoise_pair(device, dtype, steps=2, hw=32, guidance=5.0, scheduler="ddim", use_graph=True, cg_end=1.0, **extra):
    """extra: guidance_rescale / original_size / crops_coords_top_left / target_size, passed to both sides;
    callback / callback_steps to the HIP pipeline only"""
    from imagharmony_amd import schedulers as hs
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    from oracle.schedulers import EulerDiscreteScheduler as OracleEuler
    ou, hu, ocfg = build_pair(device, dtype)
    lat = det_randn((1, 4, hw, hw), 3)
    pe, ne = det_randn((1, 81, ocfg.cross_attention_dim), 4), det_randn((1, 81, ocfg.cross_attention_dim), 5)
    po, no = det_randn((1, ocfg.pooled_dim), 6), det_randn((1, ocfg.pooled_dim), 7)
    osch = OracleDDIM() if scheduler == "ddim" else OracleEuler()
    ref = oracle_denoise(ou, osch, lat, pe, ne, po, no, hw * 8, hw * 8, num_inference_steps=steps,
                         guidance_scale=guidance, control_guidance_end=cg_end,
                         **{k: v for k, v in extra.items() if not k.startswith("callback")})
    pipe = StableDiffusionXLCustomPipeline(hu, scheduler=hs.DDIMScheduler() if scheduler == "ddim" else hs.EulerDiscreteScheduler(),
                                           device=device, dtype=dtype, use_graph=use_graph)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=po, negative_pooled_prompt_embeds=no,
               height=hw * 8, width=hw * 8, num_inference_steps=steps, guidance_scale=guidance, latents=lat,
               control_guidance_end=cg_end, output_type="latent", **extra).images
    return out.float().cpu(), ref


def run_smoke(device):
    """fp16 is the sharp check (its rounding noise is 8x finer than bf16's: any indexing / layout / schedule error shows up
    far above the bound); the bf16 run -- the benchmark dtype -- is held to the bf16 noise floor of this 2-step, CFG-5
    trajectory (weights AND activations rounded to 8 bits, the cond / uncond difference amplified 5x), the same bound
    as tests/test_gpu_pipeline.py's 3-step trajectories"""
    out16, ref = denoise_pair(device, torch.float16, steps=2)
    r16 = rel_rms(out16, ref)
    out, ref = denoise_pair(device, torch.bfloat16, steps=2)
    r = rel_rms(out, ref)
    print(f"smoke: 2-step DDIM denoise (tiny UNet, CFG 5.0, IP tokens 4) rel-rms vs CPU oracle: fp16 {r16:.3e} (< 6e-3), "
          f"bf16 {r:.3e} (< 4e-2)")
    assert torch.isfinite(out16).all() and r16 < 6e-3, r16
    assert torch.isfinite(out).all() and r < 4e-2, r
