"""GPU parity of the denoise loop (device-resident step counter, hipGraph replay, CFG, scheduler tables,
IP-scale gating) and of the conditioning modules / adapter API against the oracle and the golden vectors."""
import os

import pytest
import torch

from conftest import GOLDEN, rel_rms
from oracle.detfill import det_fill, det_randn
from oracle.gen_golden import HA_CFG, RES_PLUSXL, RES_TEST
from smoke_impl import build_pair, denoise_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("sched", ["ddim", "euler"])
@pytest.mark.parametrize("use_graph", [True, False])
def test_denoise_loop_matches_oracle(sched, use_graph):
    out, ref = denoise_pair(DEV, torch.bfloat16, steps=3, scheduler=sched, use_graph=use_graph)
    r = rel_rms(out, ref)
    print(f"{sched} graph={use_graph}: 3-step denoise rel-rms {r:.3e}")
    assert r < 4e-2


def test_denoise_fp16_and_gating():
    out, ref = denoise_pair(DEV, torch.float16, steps=3, cg_end=0.5)      # IP branch gated off on the last steps
    assert rel_rms(out, ref) < 1e-2
    out2, ref2 = denoise_pair(DEV, torch.float16, steps=3, cg_end=1.0)
    assert (ref - ref2).abs().max() > 1e-4 and rel_rms(out2, ref2) < 1e-2


def test_denoise_guidance_rescale_microconditioning_and_callback():
    """rescale_noise_cfg (custom_pipelines.py:351-354) as a device-side per-sample factor, SDXL micro-conditioning
    time_ids (:277-293) and the step callback (:359-363), against the oracle loop"""
    seen = []
    out, ref = denoise_pair(DEV, torch.bfloat16, steps=3, guidance=7.0, guidance_rescale=0.7, original_size=(512, 384),
                            crops_coords_top_left=(16, 32), target_size=(256, 256),
                            callback=lambda i, t, lat: seen.append((i, float(t), lat.float().abs().mean().item())),
                            callback_steps=2)
    r = rel_rms(out, ref)
    assert torch.isfinite(out).all() and r < 3e-2, r
    assert [s[0] for s in seen] == [0, 2] and seen[0][1] > seen[1][1] and all(s[2] > 0 for s in seen)
    plain, ref_plain = denoise_pair(DEV, torch.bfloat16, steps=3, guidance=7.0)
    assert rel_rms(ref, ref_plain) > 1e-2 and rel_rms(out, plain) > 1e-2        # the options really change the result


def test_denoise_non_square_ragged_token_count():
    """256x320: the deepest level has 8x10 = 80 tokens -- not a multiple of the 64-key attention tile -- so the fused
    self-attention takes its padded per-batch path (official SDXL sizes such as 1152x896 hit the same case: 36x28 = 1008)"""
    from imagharmony_amd import schedulers as hs
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    from oracle.pipeline import denoise as oracle_denoise
    from oracle.schedulers import DDIMScheduler as OracleDDIM
    dtype = torch.bfloat16
    ou, hu, ocfg = build_pair(DEV, dtype)
    cd = ocfg.cross_attention_dim
    lat = det_randn((1, 4, 32, 40), 3)
    pe, ne = det_randn((1, 81, cd), 4), det_randn((1, 81, cd), 5)
    po, no = det_randn((1, ocfg.pooled_dim), 6), det_randn((1, ocfg.pooled_dim), 7)
    ref = oracle_denoise(ou, OracleDDIM(), lat, pe, ne, po, no, 256, 320, num_inference_steps=2, guidance_scale=5.0)
    pipe = StableDiffusionXLCustomPipeline(hu, scheduler=hs.DDIMScheduler(), device=DEV, dtype=dtype)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=po, negative_pooled_prompt_embeds=no,
               height=256, width=320, num_inference_steps=2, guidance_scale=5.0, latents=lat, output_type="latent").images
    assert out.shape == (1, 4, 32, 40)
    assert rel_rms(out.float().cpu(), ref) < 3e-2
    with pytest.raises(Exception, match="multiple of 16"):          # 8x... tokens not a multiple of 16: a clear error, not garbage
        pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=po, negative_pooled_prompt_embeds=no,
             height=384, width=320, num_inference_steps=1, guidance_scale=5.0, latents=det_randn((1, 4, 48, 40), 3),
             output_type="latent")


def test_denoise_denoising_end_truncates_like_reference():
    """denoising_end (custom_pipelines.py:303-311): the loop stops below the cut-off timestep and the IP-scale gating
    window counts the truncated list; the oracle loop for this case is pinned to the reference's own loop on CPU
    (tests/test_oracle_loop_vs_reference.py)"""
    out, ref = denoise_pair(DEV, torch.bfloat16, steps=5, denoising_end=0.6, cg_end=0.7)
    assert rel_rms(out, ref) < 3e-2
    full, _ = denoise_pair(DEV, torch.bfloat16, steps=5, cg_end=0.7)
    assert rel_rms(out, full) > 1e-2


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 6e-3), (torch.bfloat16, 4e-2)])
def test_denoise_cfg_split_over_two_engines_matches_the_fused_step(dtype, tol):
    """the two-stage PNS tail shared by two ranks (DenoiseEngine cfg_role 0 / 1 + denoise_cfg_split): each engine runs the UNet on
    ITS half of the CFG pair, the halves are exchanged every step, both apply the same combine + DDIM step.  Here the two "ranks"
    are two engines on one GPU stepped in lockstep (the exchange is a local swap; over RCCL it is pns.pair_exchange's all_gather,
    covered under gloo by tests/test_pns_gloo.py): the latents of both stay bit-equal to each other and match the fused engine and
    the CPU oracle loop (custom_pipelines.py:324-363) within the trajectory bound"""
    from imagharmony_amd import lib as L
    from imagharmony_amd import schedulers as hs
    from imagharmony_amd.denoise import DenoiseEngine
    from oracle.pipeline import denoise as oracle_denoise
    from oracle.schedulers import DDIMScheduler as OracleDDIM
    steps, hw = 3, 32
    ou, hu, ocfg = build_pair(DEV, dtype)
    lat = det_randn((1, 4, hw, hw), 3)
    pe, ne = det_randn((1, 81, ocfg.cross_attention_dim), 4), det_randn((1, 81, ocfg.cross_attention_dim), 5)
    po, no = det_randn((1, ocfg.pooled_dim), 6), det_randn((1, ocfg.pooled_dim), 7)
    ref = oracle_denoise(ou, OracleDDIM(), lat, pe, ne, po, no, hw * 8, hw * 8, num_inference_steps=steps, guidance_scale=5.0)
    engs = []
    for role in (None, 0, 1):
        e = DenoiseEngine(hu, DEV, dtype, use_graph=True)
        e.set_conditioning(pe.to(DEV), ne.to(DEV), po.to(DEV), no.to(DEV), hw * 8, hw * 8, guidance_scale=5.0, cfg_role=role)
        e.set_schedule(hs.DDIMScheduler(), steps)
        engs.append(e)
    fused = engs[0].denoise(lat).float().cpu().clone()
    with pytest.raises(L.ImhError, match="denoise_cfg_split"):
        engs[1].denoise(lat)
    # lockstep emulation of the two ranks: forward halves, swap, identical tails
    a, b = engs[1], engs[2]
    for e in (a, b):
        if e.plan is None:
            e._record()
        e.st.latents.copy_(lat.to(DEV, torch.float32) * e.init_noise_sigma)
        e.eager.ew(L.EW_STEP_SET, e.st.step, i=(0, 1, 0, 0, 0, 0), descr="step=0")
    assert a.noise_pred.shape[0] == 1                     # UNet batch S, not 2S
    for _ in range(steps):
        a.plan.replay(); b.plan.replay()
        for e in (a, b):
            e.np_full[0].copy_(a.noise_pred); e.np_full[1].copy_(b.noise_pred)
            e.plan_tail.replay()
        assert torch.equal(a.st.latents, b.st.latents)
    out = a.st.latents.float().cpu()
    r_ref, r_fused = rel_rms(out, ref), rel_rms(out, fused)
    print(f"CFG-split denoise {dtype}: rel-rms vs oracle {r_ref:.3e}, vs the fused engine {r_fused:.3e}")
    assert torch.isfinite(out).all() and r_ref < tol and r_fused < tol
    # the blocking entry point refuses an engine that holds the whole pair
    with pytest.raises(L.ImhError, match="cfg_role"):
        engs[0].denoise_cfg_split(lat, lambda m: (m, m))


def test_engine_picks_an_xcd_cell_shape_and_every_shape_gives_the_same_bits():
    """DenoiseEngine measures, when it first records its plan, which XCD cell shape of the GEMM / conv tile grids this box prefers
    (imh_gemm_args.xcd through Ctx.xcd_cells).  Placement only: every shape, the engine's pick included, denoises to the
    same bits (custom_pipelines.py:324-363 is the loop being run)"""
    from imagharmony_amd import schedulers as hs
    from imagharmony_amd.denoise import DenoiseEngine
    steps, hw = 2, 32
    ou, hu, ocfg = build_pair(DEV, torch.bfloat16)
    lat = det_randn((1, 4, hw, hw), 3)
    pe, ne = det_randn((1, 81, ocfg.cross_attention_dim), 4), det_randn((1, 81, ocfg.cross_attention_dim), 5)
    po, no = det_randn((1, ocfg.pooled_dim), 6), det_randn((1, ocfg.pooled_dim), 7)
    outs = {}
    for cells in (None, 0, 2, 3, 4, 5, 13):
        e = DenoiseEngine(hu, DEV, torch.bfloat16, use_graph=True)
        if cells is not None:
            e.xcd_candidates, e.xcd_cells = (cells,), cells
        e.set_conditioning(pe.to(DEV), ne.to(DEV), po.to(DEV), no.to(DEV), hw * 8, hw * 8, guidance_scale=5.0)
        e.set_schedule(hs.DDIMScheduler(), steps)
        outs[cells] = e.denoise(lat).float().cpu().clone()
        if cells is None:
            assert e.xcd_cells in e.xcd_candidates and set(e.xcd_times_ms) == set(e.xcd_candidates)
            assert all(t > 0 for t in e.xcd_times_ms.values())
            assert e.fork().xcd_cells == e.xcd_cells           # in-flight candidates reuse the pick
            # ... and so does every later engine of this shape in the process: the pick is measured once per (device, batch, size, dtype)
            e2 = DenoiseEngine(hu, DEV, torch.bfloat16, use_graph=True)
            e2._pick_xcd_cells = lambda: (_ for _ in ()).throw(AssertionError("the XCD policy was measured twice"))
            e2.set_conditioning(pe.to(DEV), ne.to(DEV), po.to(DEV), no.to(DEV), hw * 8, hw * 8, guidance_scale=5.0)
            e2.set_schedule(hs.DDIMScheduler(), steps)
            assert torch.equal(e2.denoise(lat).float().cpu(), outs[None]) and e2.xcd_cells == e.xcd_cells
            # alternating two schedules (two-stage PNS: preview / final) re-records nothing the second time round
            p_short = e2.plan
            e2.set_schedule(hs.DDIMScheduler(), steps + 1)
            long1 = e2.denoise(lat).float().cpu().clone()
            p_long = e2.plan
            e2.set_schedule(hs.DDIMScheduler(), steps)
            assert e2.plan is p_short and torch.equal(e2.denoise(lat).float().cpu(), outs[None])
            e2.set_schedule(hs.DDIMScheduler(), steps + 1)
            assert e2.plan is p_long and torch.equal(e2.denoise(lat).float().cpu(), long1)
    for cells, o in outs.items():
        assert torch.equal(o, outs[0]), f"xcd cells {cells}"


def test_denoise_is_deterministic_and_replayable():
    a, _ = denoise_pair(DEV, torch.bfloat16, steps=2)
    b, _ = denoise_pair(DEV, torch.bfloat16, steps=2)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2e-2)])
def test_harmony_imageproj_match_reference_golden(dtype, tol):
    from imagharmony_amd.modules import HarmonyAttention, ImageProjModel
    g = torch.load(os.path.join(GOLDEN, "harmony_imageproj.pt"))
    ha = det_fill(HarmonyAttention(**HA_CFG), 23, prefix="ha.").to(DEV, dtype)
    text, img = det_randn((1, 77, 2048), 31).to(DEV, dtype), det_randn((1, 1280), 32).to(DEV, dtype)
    out = ha(text, img)
    assert rel_rms(out.float().cpu(), g["ha_out"]) < tol
    proj = det_fill(ImageProjModel(2048, 1280, 4), 29, prefix="proj.").to(DEV, dtype)
    fused = (det_randn((1, 1280), 32) + g["ha_out"]).to(DEV, dtype)
    assert rel_rms(proj(fused).float().cpu(), g["tokens"]) < tol
    assert rel_rms(proj(torch.zeros_like(fused)).float().cpu(), g["uncond_tokens"]) < tol


@pytest.mark.parametrize("name,cfg", [("plusxl", RES_PLUSXL), ("testcfg", RES_TEST)])
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_resampler_matches_reference_golden(name, cfg, dtype, tol):
    from imagharmony_amd.modules import Resampler
    g = torch.load(os.path.join(GOLDEN, f"resampler_{name}.pt"))
    r = det_fill(Resampler(**cfg), 37, prefix="res.").to(DEV, dtype)
    y = r(det_randn((g["batch"], 257, cfg["embedding_dim"]), 41).to(DEV, dtype))
    assert y.shape == g["out"].shape                                   # the reference's only test (test_resampler.py:40)
    assert rel_rms(y.float().cpu(), g["out"]) < tol


def test_ipadapterxl_generate_call_sequence():
    """test.py's call sequence on the reduced config with injected CLIP embeddings / prompt embeddings
    (no encoders offline): IPAdapterXL(...) -> generate(...) -> latents; seeds reproduce, scale matters."""
    from imagharmony_amd.ip_adapter import IPAdapterXL
    from imagharmony_amd.modules import HarmonyAttention
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    dtype = torch.float16
    ou, hu, ocfg = build_pair(DEV, dtype)
    pipe = StableDiffusionXLCustomPipeline(hu, device=DEV, dtype=dtype)
    ha = det_fill(HarmonyAttention(image_hidden_size=128, text_context_dim=ocfg.cross_attention_dim, inter_dim=512,
                                   cross_heads=8, reshape_blocks=8, cross_value_dim=64), 3)
    ip = IPAdapterXL(pipe, None, None, DEV, num_tokens=4, inference=True, number_class_crossattention=ha,
                     dtype=dtype, clip_embeddings_dim=128)
    det_fill(ip.image_proj_model, 5)
    for n, p in hu.attn_processors.items():
        det_fill(p, 9, prefix=n)
    cd = ocfg.cross_attention_dim
    embeds = (det_randn((1, 77, cd), 1), det_randn((1, 77, cd), 2), det_randn((1, ocfg.pooled_dim), 3),
              det_randn((1, ocfg.pooled_dim), 4))
    kw = dict(clip_image_embeds=det_randn((1, 128), 5), prompt_embeds=embeds, extra_prompt_embeds=det_randn((1, 77, cd), 6),
              num_samples=1, num_inference_steps=2, guidance_scale=5.0, height=256, width=256,
              number_class_crossattention=ha, output_type="latent")
    a = ip.generate(seed=42, scale=1.0, **kw)
    b = ip.generate(seed=42, scale=1.0, **kw)
    c = ip.generate(seed=42, scale=0.0, **kw)
    d = ip.generate(seed=43, scale=1.0, **kw)
    assert a.shape == (1, 4, 32, 32) and torch.isfinite(a).all()
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)


def test_end_to_end_pil_in_pil_out_like_test_py():
    """test.py's whole flow on reduced configs, nothing injected except prompt embeddings (no tokenizer vocabulary
    offline): PIL image -> CLIPImageProcessor -> CLIP vision model (transformers, random weights; the step before
    the path) -> HarmonyAttention + ImageProjModel -> IP tokens -> denoise -> VAE tiled decode ->
    post-processing -> PIL image."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModelWithProjection
    from imagharmony_amd.ip_adapter import IPAdapterPlusXL, IPAdapterXL
    from imagharmony_amd.modules import HarmonyAttention
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    from imagharmony_amd.vae import AutoencoderKL, VAEConfig
    dtype = torch.bfloat16
    ou, hu, ocfg = build_pair(DEV, dtype)
    vae = AutoencoderKL(VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1, sample_size=256)).init_random_(2).to(DEV, dtype)
    pipe = StableDiffusionXLCustomPipeline(hu, device=DEV, dtype=dtype, vae=vae)
    pipe.enable_vae_tiling()
    torch.manual_seed(0)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                                          num_attention_heads=4, image_size=32, patch_size=8,
                                                          projection_dim=128)).eval().to(DEV, dtype)
    proc = CLIPImageProcessor(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32})
    img = Image.fromarray((np.random.RandomState(0).rand(48, 40, 3) * 255).astype("uint8"))
    cd = ocfg.cross_attention_dim
    ha = det_fill(HarmonyAttention(image_hidden_size=128, text_context_dim=cd, inter_dim=512, cross_heads=8,
                                   reshape_blocks=8, cross_value_dim=64), 3)
    ip = IPAdapterXL(pipe, None, None, DEV, num_tokens=4, inference=True, number_class_crossattention=ha, dtype=dtype,
                     image_encoder=clip, clip_image_processor=proc)
    assert ip.clip_embeddings_dim == 128                       # read from the encoder's config (ip_adapter.py:93-95)
    det_fill(ip.image_proj_model, 5)
    embeds = (det_randn((1, 77, cd), 1), det_randn((1, 77, cd), 2), det_randn((1, ocfg.pooled_dim), 3),
              det_randn((1, ocfg.pooled_dim), 4))
    out = ip.generate(pil_image=img, prompt_embeds=embeds, extra_prompt_embeds=det_randn((1, 77, cd), 6), num_samples=1,
                      seed=42, num_inference_steps=2, guidance_scale=5.0, height=256, width=320, output_type="pil")
    assert isinstance(out[0], Image.Image) and out[0].size == (320, 256)
    # --- the reference's own call, verbatim (test.py:28-43): string prompts, stray number_class_crossattention=,
    #     NO output_type (the reference pipeline defaults to "pil", custom_pipelines.py:42), then images[0].save() ---
    def fake_text_encoder(prompt, num_images_per_prompt=1, do_classifier_free_guidance=True, negative_prompt=None):
        # no tokenizer vocabulary offline: deterministic embeddings keyed by the strings (the step before the path)
        def emb(txt, n, d):
            sd = sum(txt.encode()) if isinstance(txt, str) else sum(sum(t.encode()) for t in txt)
            return det_randn((num_images_per_prompt, n, d) if n else (num_images_per_prompt, d), sd % 1000)
        return (emb(prompt, 77, cd), emb(negative_prompt or "", 77, cd), emb(prompt, 0, ocfg.pooled_dim),
                emb(negative_prompt or "", 0, ocfg.pooled_dim))
    pipe.text_encoder = fake_text_encoder
    ip_model, input_image, number_class_crossattention = ip, img.resize((512, 512)), ha
    prompt, extra_text = "a photo of three cats", "three cats"
    images = ip_model.generate(
        pil_image=input_image,
        prompt=prompt,
        negative_prompt="text, watermark, lowres, low quality, worst quality, deformed, glitch, low contrast, noisy, saturation, blurry",
        scale=1.0,
        guidance_scale=5.0,
        num_samples=1,
        num_inference_steps=30,
        seed=42,
        extra_text=extra_text,
        number_class_crossattention=number_class_crossattention
    )
    import tempfile, os
    with tempfile.TemporaryDirectory() as td:
        output_path = os.path.join(td, "output.png")
        images[0].save(output_path)
        assert Image.open(output_path).size == (256, 256)       # default_sample_size 32 x vae_scale_factor 8
    plus = IPAdapterPlusXL(pipe, None, None, DEV, num_tokens=16, dtype=dtype, image_encoder=clip, clip_image_processor=proc)
    assert plus.clip_hidden_size == 64
    a, b = plus.get_image_embeds(pil_image=img)                # penultimate hidden states -> Resampler (ip_adapter.py:405-417)
    assert a.shape == b.shape == (1, 16, cd) and torch.isfinite(a.float()).all() and not torch.equal(a, b)


def test_pns_single_rank_on_device():
    """PNS driver end-to-end on one GPU: 3 candidate seeds through the HIP engine (2-step preview), winner re-denoised
    with more steps; the selection must be reproducible and the winner's latent must equal a direct denoise of that seed."""
    from imagharmony_amd import pns
    from imagharmony_amd import schedulers as hs
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    dtype = torch.bfloat16
    ou, hu, ocfg = build_pair(DEV, dtype)
    pipe = StableDiffusionXLCustomPipeline(hu, scheduler=hs.DDIMScheduler(), device=DEV, dtype=dtype)
    cd = ocfg.cross_attention_dim
    pe, ne = det_randn((1, 81, cd), 4), det_randn((1, 81, cd), 5)
    po, no = det_randn((1, ocfg.pooled_dim), 6), det_randn((1, ocfg.pooled_dim), 7)
    eng = pipe.engine
    eng.set_conditioning(pe, ne, po, no, 256, 256, guidance_scale=5.0)

    def preview(noise):
        eng.set_schedule(pipe.scheduler, 2)
        return eng.denoise(noise).clone()

    def final(noise):
        eng.set_schedule(pipe.scheduler, 3)
        return eng.denoise(noise).clone()

    r1 = pns.run_pns(preview, [3, 9, 27], (1, 4, 32, 32), device=DEV, final_fn=final)
    r2 = pns.run_pns(preview, [3, 9, 27], (1, 4, 32, 32), device=DEV, final_fn=final)
    assert r1["best_seed"] == r2["best_seed"] and torch.equal(r1["latents"], r2["latents"])
    assert torch.equal(r1["scores"], r2["scores"]) and torch.isfinite(r1["scores"]).all()
    direct = final(pns.seed_latents(r1["best_seed"], (1, 4, 32, 32)))
    assert torch.equal(direct, r1["latents"])


def test_pns_candidates_batched_per_forward():
    """configs[4]-style PNS: several candidate seeds of one rank stacked into one UNet batch (S=3 -> UNet batch 6).
    Candidates are independent rows of every op, so each one matches its own batch-1 denoise to rounding and the
    selection is unchanged."""
    from imagharmony_amd import pns
    from imagharmony_amd import schedulers as hs
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    dtype = torch.bfloat16
    ou, hu, ocfg = build_pair(DEV, dtype)
    pipe = StableDiffusionXLCustomPipeline(hu, scheduler=hs.DDIMScheduler(), device=DEV, dtype=dtype)
    cd = ocfg.cross_attention_dim
    pe, ne = det_randn((1, 81, cd), 4), det_randn((1, 81, cd), 5)
    po, no = det_randn((1, ocfg.pooled_dim), 6), det_randn((1, ocfg.pooled_dim), 7)
    seeds = [3, 9, 27]

    def engine_for(S):
        eng = pipe.engine if S == 1 else pipe.engine.__class__(hu, DEV, dtype, True)
        eng.set_conditioning(pe.repeat(S, 1, 1), ne.repeat(S, 1, 1), po.repeat(S, 1), no.repeat(S, 1), 256, 256,
                             guidance_scale=5.0)
        eng.set_schedule(pipe.scheduler, 2)
        return eng

    e1, e3 = engine_for(1), engine_for(3)
    one = pns.run_pns(lambda z: e1.denoise(z).clone(), seeds, (1, 4, 32, 32), device=DEV)
    stacked = pns.run_pns(lambda z: e3.denoise(z).clone(), seeds, (1, 4, 32, 32), device=DEV, batch=3)
    assert stacked["best_seed"] == one["best_seed"]
    assert torch.allclose(stacked["scores"], one["scores"], atol=2e-2)
    assert rel_rms(stacked["latents"].cpu(), one["latents"].cpu()) < 2e-2


def test_hip_denoise_matches_committed_oracle_fixture():
    """the HIP engine against tests/golden/oracle_tiny_unet.pt (a committed oracle trajectory: the GPU box has no
    /root/reference and this also pins HIP vs oracle without recomputing the oracle)"""
    g = torch.load(os.path.join(GOLDEN, "oracle_tiny_unet.pt"))
    out, _ = denoise_pair(DEV, torch.bfloat16, steps=2)
    assert rel_rms(out, g["ddim_step2"].float()) < 3e-2


def test_ipadapter_plus_xl_generate_call_sequence():
    """IPAdapterPlusXL (Resampler over penultimate CLIP hidden states, ip_adapter.py:389-478) on the reduced config
    with injected hidden states: 16 image tokens flow through the decoupled cross-attention."""
    from imagharmony_amd.ip_adapter import IPAdapterPlusXL
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    dtype = torch.float16
    ou, hu, ocfg = build_pair(DEV, dtype, num_tokens=16)
    pipe = StableDiffusionXLCustomPipeline(hu, device=DEV, dtype=dtype)
    ip = IPAdapterPlusXL(pipe, None, None, DEV, num_tokens=16, dtype=dtype, clip_hidden_size=128)
    det_fill(ip.image_proj_model, 5)
    for n, p in hu.attn_processors.items():
        det_fill(p, 9, prefix=n)
    cd = ocfg.cross_attention_dim
    embeds = (det_randn((1, 77, cd), 1), det_randn((1, 77, cd), 2), det_randn((1, ocfg.pooled_dim), 3),
              det_randn((1, ocfg.pooled_dim), 4))
    kw = dict(clip_hidden_states=det_randn((1, 257, 128), 5), uncond_clip_hidden_states=det_randn((1, 257, 128), 6),
              prompt_embeds=embeds, num_samples=1, num_inference_steps=2, guidance_scale=5.0, height=256, width=256,
              output_type="latent")
    a = ip.generate(seed=7, scale=1.0, **kw)
    b = ip.generate(seed=7, scale=1.0, **kw)
    c = ip.generate(seed=7, scale=0.3, **kw)
    assert a.shape == (1, 4, 32, 32) and torch.isfinite(a).all()
    assert ip.image_proj_model(kw["clip_hidden_states"].to(DEV, dtype)).shape == (1, 16, cd)
    assert torch.equal(a, b) and not torch.equal(a, c)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2e-2)])
def test_mlpproj_matches_reference_golden(dtype, tol):
    """MLPProjModel (IPAdapterFull, ip_adapter.py:50-66) on the HIP ops (GELU in the GEMM epilogue)"""
    from imagharmony_amd.modules import MLPProjModel
    from oracle.gen_golden import MLP_CFG
    g = torch.load(os.path.join(GOLDEN, "mlpproj.pt"))
    m = det_fill(MLPProjModel(**MLP_CFG), 43, prefix="mlp.").to(DEV, dtype)
    y = m(det_randn((1, 64, MLP_CFG["clip_embeddings_dim"]), 47).to(DEV, dtype))
    assert y.shape == g["out"].shape and rel_rms(y.float().cpu(), g["out"]) < tol


def test_ipadapter_plus_and_full_surface():
    """IPAdapterPlus / IPAdapterFull (ip_adapter.py:344-386): projection modules, state-dict keys and the base
    generate() call sequence with injected CLIP hidden states (SD-1.x style pipe: 2-tuple encode_prompt)."""
    import imagharmony_amd as pkg
    from imagharmony_amd.modules import MLPProjModel, Resampler

    class _Cfg:
        cross_attention_dim = 256
        block_out_channels = (64, 128, 256)

    class _Pipe:
        def __init__(self, unet):
            self.unet = unet
            self.calls = []

        def to(self, device):
            return self

        def __call__(self, **kw):
            self.calls.append(kw)
            return type("O", (), {"images": ["img"] * kw["prompt_embeds"].shape[0]})()

    dtype = torch.bfloat16
    ou, hu, ocfg = build_pair(DEV, dtype)
    pipe = _Pipe(hu)
    plus = pkg.IPAdapterPlus(pipe, None, None, DEV, num_tokens=16, dtype=dtype, clip_hidden_size=128)
    full = pkg.IPAdapterFull(pipe, None, None, DEV, num_tokens=65, dtype=dtype, clip_hidden_size=128)
    assert isinstance(plus.image_proj_model, Resampler) and isinstance(full.image_proj_model, MLPProjModel)
    assert list(full.image_proj_model.state_dict()) == ["proj.0.weight", "proj.0.bias", "proj.2.weight", "proj.2.bias",
                                                        "proj.3.weight", "proj.3.bias"]
    hid, hid0 = det_randn((1, 65, 128), 9).to(DEV, dtype), det_randn((1, 65, 128), 10).to(DEV, dtype)
    a, b = plus.get_image_embeds(clip_image_embeds=hid, uncond_clip_image_embeds=hid0)
    assert a.shape == b.shape == (1, 16, ocfg.cross_attention_dim) and torch.isfinite(a.float()).all()
    pe, ne = det_randn((2, 77, ocfg.cross_attention_dim), 11).to(DEV, dtype), det_randn((2, 77, ocfg.cross_attention_dim), 12).to(DEV, dtype)
    imgs = full.generate(clip_image_embeds=hid, uncond_clip_image_embeds=hid0, num_samples=2, seed=7, guidance_scale=7.5,
                         num_inference_steps=3, prompt_embeds=(pe, ne))
    kw = pipe.calls[-1]
    assert len(imgs) == 2 and kw["prompt_embeds"].shape == (2, 77 + 65, ocfg.cross_attention_dim)
    assert kw["guidance_scale"] == 7.5 and kw["num_inference_steps"] == 3 and "pooled_prompt_embeds" not in kw


def test_generate_pns_two_stage_with_clip_judge():
    """PNS as the reference's figure draws it (README.md:27, assets/1.png): candidate seeds -> 10-step previews ->
    judge -> best noise -> 30-step final, here with the default CLIP-space judge (decoded preview vs the
    harmony-fused image embedding) on reduced configs; deterministic, and the final image is the 30-step denoise of
    the winning seed."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModelWithProjection
    from imagharmony_amd import pns
    from imagharmony_amd.ip_adapter import IPAdapterXL
    from imagharmony_amd.modules import HarmonyAttention
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    from imagharmony_amd.vae import AutoencoderKL, VAEConfig
    dtype = torch.bfloat16
    ou, hu, ocfg = build_pair(DEV, dtype)
    vae = AutoencoderKL(VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1, sample_size=256)).init_random_(2).to(DEV, dtype)
    pipe = StableDiffusionXLCustomPipeline(hu, device=DEV, dtype=dtype, vae=vae)
    torch.manual_seed(0)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                                          num_attention_heads=4, image_size=32, patch_size=8,
                                                          projection_dim=128)).eval().to(DEV, dtype)
    proc = CLIPImageProcessor(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32})
    img = Image.fromarray((np.random.RandomState(0).rand(48, 40, 3) * 255).astype("uint8"))
    cd = ocfg.cross_attention_dim
    ha = det_fill(HarmonyAttention(image_hidden_size=128, text_context_dim=cd, inter_dim=512, cross_heads=8,
                                   reshape_blocks=8, cross_value_dim=64), 3)
    ip = IPAdapterXL(pipe, None, None, DEV, num_tokens=4, inference=True, number_class_crossattention=ha, dtype=dtype,
                     image_encoder=clip, clip_image_processor=proc)
    det_fill(ip.image_proj_model, 5)
    embeds = (det_randn((1, 77, cd), 1), det_randn((1, 77, cd), 2), det_randn((1, ocfg.pooled_dim), 3),
              det_randn((1, ocfg.pooled_dim), 4))
    kw = dict(pil_image=img, prompt_embeds=embeds, extra_prompt_embeds=det_randn((1, 77, cd), 6), preview_steps=10,
              num_inference_steps=30, guidance_scale=5.0, height=256, width=256)
    seeds = [3, 9, 27, 81]
    r1 = ip.generate_pns(seeds, batch=1, **kw)               # one candidate per forward (the rounds 3-5 default)
    r2 = ip.generate_pns(seeds, batch=2, **kw)               # two candidates stacked per forward: same winner
    r4 = ip.generate_pns(seeds, **kw)                        # the default: all four of this rank's seeds in one UNet batch of 8
    assert r4["best_seed"] == r1["best_seed"] or max(r1["scores"].flatten().tolist()) - r1["scores"].flatten().tolist()[seeds.index(r4["best_seed"])] < 5e-2
    assert (r1["scores"] - r4["scores"]).abs().max() < 5e-2
    assert isinstance(r1["images"][0], Image.Image) and r1["images"][0].size == (256, 256)
    assert r1["best_seed"] in seeds and r2["best_seed"] in seeds
    # same winner -- unless two candidates tie within the batched-vs-batch-1 rounding band allowed below (random CLIP
    # and random weights put the four scores close together)
    s1 = r1["scores"].flatten().tolist()
    assert r1["best_seed"] == r2["best_seed"] or max(s1) - s1[seeds.index(r2["best_seed"])] < 5e-2
    assert torch.isfinite(r1["scores"]).all() and r1["scores"].abs().max() <= 1.0 + 1e-3
    assert r1["scores"].unique().numel() == len(seeds)
    assert (r1["scores"] - r2["scores"]).abs().max() < 5e-2      # batched rows differ from batch-1 rows by rounding only
    # the returned latent is the 30-step denoise of the winning noise
    direct = ip.generate_pns([r1["best_seed"]], output_type="latent", **kw)["latents"]
    assert torch.equal(direct, r1["latents"])


def test_string_prompts_through_gpu_clip_text_encoders():
    """SURVEY.md 8f-4 on the device: two stock transformers CLIP text encoders (random weights, stub tokenizers -- no
    vocabulary offline) on the GPU behind imagharmony_amd.text.SDXLPromptEncoder feed IPAdapterXL.generate with STRING
    prompts (ip_adapter.py:285-319 -> encode_prompt); the result equals the same run with the encoder's embeddings
    passed in explicitly, and changes with the prompt."""
    import numpy as np
    from PIL import Image
    from transformers import (CLIPImageProcessor, CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection, CLIPVisionConfig,
                              CLIPVisionModelWithProjection)
    from imagharmony_amd.ip_adapter import IPAdapterXL
    from imagharmony_amd.modules import HarmonyAttention
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    from imagharmony_amd.text import SDXLPromptEncoder
    from tests.test_text_encoder import _Tok
    dtype = torch.bfloat16
    ou, hu, ocfg = build_pair(DEV, dtype)
    cd, pd = ocfg.cross_attention_dim, ocfg.pooled_dim

    class Tok77(_Tok):
        model_max_length = 77
    torch.manual_seed(0)
    h1 = cd // 4
    c1 = CLIPTextConfig(vocab_size=100, hidden_size=h1, intermediate_size=2 * h1, num_hidden_layers=2, num_attention_heads=4,
                        max_position_embeddings=77, projection_dim=h1)
    c2 = CLIPTextConfig(vocab_size=100, hidden_size=cd - h1, intermediate_size=cd, num_hidden_layers=2, num_attention_heads=4,
                        max_position_embeddings=77, projection_dim=pd)
    e1, e2 = CLIPTextModel(c1).eval().to(DEV, dtype), CLIPTextModelWithProjection(c2).eval().to(DEV, dtype)
    enc = SDXLPromptEncoder(Tok77(), Tok77(), e1, e2)
    pipe = StableDiffusionXLCustomPipeline(hu, device=DEV, dtype=dtype, text_encoder=enc)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                                          num_attention_heads=4, image_size=32, patch_size=8,
                                                          projection_dim=128)).eval().to(DEV, dtype)
    proc = CLIPImageProcessor(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32})
    img = Image.fromarray((np.random.RandomState(0).rand(48, 40, 3) * 255).astype("uint8"))
    ha = det_fill(HarmonyAttention(image_hidden_size=128, text_context_dim=cd, inter_dim=512, cross_heads=8,
                                   reshape_blocks=8, cross_value_dim=64), 3)
    ip = IPAdapterXL(pipe, None, None, DEV, num_tokens=4, inference=True, number_class_crossattention=ha, dtype=dtype,
                     image_encoder=clip, clip_image_processor=proc)
    det_fill(ip.image_proj_model, 5)
    kw = dict(pil_image=img, num_samples=1, seed=7, num_inference_steps=2, guidance_scale=5.0, height=256, width=256,
              output_type="latent")
    a = ip.generate(prompt="a photo of three cats", negative_prompt="blurry", extra_text="three cats", **kw)
    b = ip.generate(prompt="a photo of two dogs", negative_prompt="blurry", extra_text="two dogs", **kw)
    assert a.shape == (1, 4, 32, 32) and torch.isfinite(a).all() and not torch.equal(a, b)
    pe, ne, pp, npp = enc("a photo of three cats", negative_prompt="blurry")
    assert pe.shape == (1, 77, cd) and pp.shape == (1, pd) and pe.is_cuda
    ex = enc("three cats", do_classifier_free_guidance=False)[0]
    c = ip.generate(prompt_embeds=(pe, ne, pp, npp), extra_prompt_embeds=ex, **kw)
    assert torch.equal(a, c)
