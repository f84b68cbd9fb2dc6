"""Mint tests/golden/*.pt by executing the reference's OWN modules (verbatim import,
oracle/refshim.py) on CPU in fp32.  Run in the build container only:

    python -m oracle.gen_golden

Each fixture stores the reference OUTPUT plus the recipe (seed, shapes, config) needed
to regenerate inputs / weights through oracle.detfill -- weights are never committed.
The reference has no golden vectors of its own; these are it.
"""
import contextlib
import io
import os

import torch

from . import refshim
from .detfill import det_fill, det_randn
from .sdxl_unet import Attention

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

ATTN_CASES = {
    # name: (B, L, C, heads, cross_dim, n_text, T_ip, scale)
    "c640_t4": (2, 96, 640, 10, 2048, 77, 4, 0.7),
    "c1280_t4": (1, 64, 1280, 20, 2048, 77, 4, 1.0),
    "c1280_t16": (2, 32, 1280, 20, 2048, 77, 16, 0.5),
    "c128_t32": (2, 160, 128, 2, 256, 77, 32, 1.0),
}


# The shapes the benchmarked forward actually runs (cfg2 = 1024^2, CFG batch 2; cfg4 = batch 8, 16 Resampler
# tokens).  The full outputs are 5-42 MB each, so a fixture keeps (a) SAMPLE_ROWS seeded token rows per batch element in
# fp16 and (b) fp32 row sums [B, L] and column sums [B, C] of the WHOLE output: every token and every channel of the
# reference result is pinned, the sampled rows element by element.
ATTN_CFG_CASES = {
    # name: (B, L, C, heads, cross_dim, n_text, T_ip, scale)
    "cfg2_c1280_L1024_t4": (2, 1024, 1280, 20, 2048, 77, 4, 1.0),
    "cfg2_c640_L4096_t4": (2, 4096, 640, 10, 2048, 77, 4, 1.0),
    "cfg4_c1280_L1024_t16_b8": (8, 1024, 1280, 20, 2048, 77, 16, 1.0),
}
SAMPLE_ROWS = 48


def all_cases():
    return {**ATTN_CASES, **ATTN_CFG_CASES}


def sample_rows(case):
    """the seeded token rows (sorted, same for every batch element) a cfg-shape fixture stores in full"""
    b, l = all_cases()[case][:2]
    g = torch.Generator(device="cpu").manual_seed(977 + l)
    return torch.randperm(l, generator=g)[:SAMPLE_ROWS].sort().values


def compress(case, y):
    """[B, L, C] fp32 reference output -> what the fixture keeps"""
    idx = sample_rows(case)
    return {"rows": y[:, idx].to(torch.float16), "row_sum": y.double().sum(-1).float(), "col_sum": y.double().sum(1).float(),
            "rms": float(y.double().pow(2).mean().sqrt())}


def attn_inputs(case, seed=11):
    b, l, c, h, cd, nt, t, scale = all_cases()[case]
    hs = det_randn((b, l, c), seed + 1)
    ehs = det_randn((b, nt + t, cd), seed + 2)
    return hs, ehs


def make_attn(case, cross, seed=11):
    b, l, c, h, cd, nt, t, scale = all_cases()[case]
    a = Attention(c, h, 64, cross_attention_dim=cd if cross else None)
    return det_fill(a, seed + 3, prefix="attn.")


HA_CFG = dict(image_hidden_size=1280, text_context_dim=2048, inter_dim=2560, cross_heads=8,
              reshape_blocks=8, cross_value_dim=64, scale=1.0, fusion_method="cross_attention")   # test.py:12-15,82-91
RES_PLUSXL = dict(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, embedding_dim=1280,
                  output_dim=2048, ff_mult=4)                                                     # ip_adapter.py:392-403
RES_TEST = dict(dim=1024, depth=2, dim_head=64, heads=16, num_queries=8, embedding_dim=1280,
                output_dim=1280, ff_mult=2, max_seq_len=257, apply_pos_emb=True,
                num_latents_mean_pooled=4)                                                        # test_resampler.py:18-30


@torch.no_grad()
def main():
    ref = refshim.load()
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(False)

    # ---- attention processors (attention_processor.py:244-465) ----
    for case, (b, l, c, h, cd, nt, t, scale) in ATTN_CASES.items():
        hs, ehs = attn_inputs(case)
        attn = make_attn(case, cross=True)
        out = {}
        for skip in (False, True):
            p = ref.IPAttnProcessor2_0(c, cd, scale=scale, num_tokens=t, skip=skip)
            det_fill(p, 17, prefix="proc.")
            out[f"ip_skip{int(skip)}"] = p(attn, hs, encoder_hidden_states=ehs).clone()
            if not skip:
                out["attn_map"] = p.attn_map.clone()
        # ControlNet processor == skip=True path (attention_processor.py:469-621)
        out["cn"] = ref.CNAttnProcessor2_0(num_tokens=t)(attn, hs, encoder_hidden_states=ehs).clone()
        sattn = make_attn(case, cross=False)
        out["self"] = ref.AttnProcessor2_0()(sattn, hs).clone()
        torch.save({"case": case, "cfg": ATTN_CASES[case], **out}, os.path.join(OUT, f"attn_{case}.pt"))
        print(case, {k: tuple(v.shape) for k, v in out.items()})

    cfg_shape_fixtures(ref)

    # ---- HarmonyAttention + ImageProjModel (train.py:188-266, ip_adapter.py:28-48, :170-176) ----
    with contextlib.redirect_stdout(io.StringIO()):      # reference prints in ctor/forward
        ha = ref.HarmonyAttention(**HA_CFG)
        det_fill(ha, 23, prefix="ha.")
        text = det_randn((1, 77, 2048), 31)
        img = det_randn((1, 1280), 32)
        ha_out = ha(text, img)
    proj = ref.ImageProjModel(cross_attention_dim=2048, clip_embeddings_dim=1280, clip_extra_context_tokens=4)
    det_fill(proj, 29, prefix="proj.")
    fused = img + ha_out
    torch.save({"ha_out": ha_out, "tokens": proj(fused), "uncond_tokens": proj(torch.zeros_like(fused)),
                "ha_cfg": HA_CFG}, os.path.join(OUT, "harmony_imageproj.pt"))
    print("harmony", tuple(ha_out.shape))

    # ---- Resampler (resampler.py:81-147) ----
    for name, cfg, bsz in (("plusxl", RES_PLUSXL, 2), ("testcfg", RES_TEST, 2)):
        r = ref.Resampler(**cfg)
        det_fill(r, 37, prefix="res.")
        x = det_randn((bsz, 257, cfg["embedding_dim"]), 41)
        y = r(x)
        torch.save({"out": y, "cfg": cfg, "batch": bsz}, os.path.join(OUT, f"resampler_{name}.pt"))
        print("resampler", name, tuple(y.shape))


@torch.no_grad()
def cfg_shape_fixtures(ref=None):
    """a1-a3 at the shapes of the benchmarked forward, produced by the VERBATIM reference classes
    (ip_adapter/attention_processor.py:258-332, :364-465)"""
    ref = ref or refshim.load()
    for case, (b, l, c, h, cd, nt, t, scale) in ATTN_CFG_CASES.items():
        hs, ehs = attn_inputs(case)
        attn = make_attn(case, cross=True)
        out = {}
        for skip in (False, True):
            p = ref.IPAttnProcessor2_0(c, cd, scale=scale, num_tokens=t, skip=skip)
            det_fill(p, 17, prefix="proc.")
            out[f"ip_skip{int(skip)}"] = compress(case, p(attn, hs, encoder_hidden_states=ehs))
            if hasattr(p, "attn_map"):
                del p.attn_map
        sattn = make_attn(case, cross=False)
        out["self"] = compress(case, ref.AttnProcessor2_0()(sattn, hs))
        torch.save({"case": case, "cfg": ATTN_CFG_CASES[case], "sample_rows": sample_rows(case), **out},
                   os.path.join(OUT, f"attn_{case}.pt"))
        print(case, {k: tuple(v["rows"].shape) for k, v in out.items()})


MLP_CFG = dict(cross_attention_dim=768, clip_embeddings_dim=1280)     # IPAdapterFull on an SD-1.x UNet + ViT-H hidden states


@torch.no_grad()
def mlpproj_fixture():
    """MLPProjModel (ip_adapter.py:50-66) run by the REFERENCE class on seeded weights / inputs."""
    ref = refshim.load()
    m = ref.MLPProjModel(**MLP_CFG)
    det_fill(m, 43, prefix="mlp.")
    x = det_randn((1, 64, MLP_CFG["clip_embeddings_dim"]), 47)       # 64 of the 257 hidden-state tokens: small fixture
    torch.save({"out": m(x), "cfg": MLP_CFG}, os.path.join(OUT, "mlpproj.pt"))
    print("mlpproj", tuple(m(x).shape))


def unet_fixture():
    """Reduced-width UNet forward + 2-step DDIM/CFG trajectory produced by the ORACLE itself (diffusers cannot be
    executed here, so this pins the oracle against drift, not against the reference)."""
    from .pipeline import denoise, install_ip_processors
    from .schedulers import DDIMScheduler
    from .sdxl_unet import UNet2DConditionModel, tiny_config
    from . import modules as om
    cfg = tiny_config()
    u = det_fill(UNet2DConditionModel(cfg), 5).eval()
    procs = install_ip_processors(u, num_tokens=4, scale=0.8)
    for n, p in procs.items():
        if isinstance(p, om.IPAttnProcessor2_0):
            det_fill(p, 7, prefix=n)
    x, ehs = det_randn((2, 4, 32, 32), 3), det_randn((2, 81, cfg.cross_attention_dim), 4)
    te = det_randn((2, cfg.pooled_dim), 6)
    ids = torch.tensor([[256, 256, 0, 0, 256, 256]], dtype=torch.float32).repeat(2, 1)
    fwd = u(x, torch.tensor(500.0), ehs, added_cond_kwargs={"text_embeds": te, "time_ids": ids})[0]
    lat = det_randn((1, 4, 32, 32), 3)
    pe, ne = det_randn((1, 81, cfg.cross_attention_dim), 4), det_randn((1, 81, cfg.cross_attention_dim), 5)
    po, no = det_randn((1, cfg.pooled_dim), 6), det_randn((1, cfg.pooled_dim), 7)
    traj = []
    denoise(u, DDIMScheduler(), lat, pe, ne, po, no, 256, 256, num_inference_steps=2, guidance_scale=5.0, trace=traj)
    return {"forward": fwd.half(), "ddim_step1": traj[0].half(), "ddim_step2": traj[1].half()}


if __name__ == "__main__":
    main()
    with torch.no_grad():
        torch.save(unet_fixture(), os.path.join(OUT, "oracle_tiny_unet.pt"))
    print("oracle_tiny_unet.pt written")
