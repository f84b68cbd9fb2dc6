"""CPU: the oracle restatement (oracle/modules.py) against the golden vectors minted from
the reference's own modules (oracle/gen_golden.py), and -- when /root/reference is present --
against the verbatim reference modules live."""
import contextlib
import io
import os

import pytest
import torch

from oracle import modules as om
from oracle import refshim
from oracle.detfill import det_fill, det_randn
from oracle.gen_golden import ATTN_CASES, HA_CFG, RES_PLUSXL, RES_TEST, attn_inputs, make_attn
from conftest import GOLDEN, rel_rms

TOL = 2e-6   # fp32 vs fp32, same math, different op order


@pytest.mark.parametrize("case", list(ATTN_CASES))
def test_attn_processors_match_reference_golden(case):
    g = torch.load(os.path.join(GOLDEN, f"attn_{case}.pt"))
    b, l, c, h, cd, nt, t, scale = ATTN_CASES[case]
    hs, ehs = attn_inputs(case)
    attn = make_attn(case, cross=True)
    with torch.no_grad():
        for skip in (False, True):
            p = det_fill(om.IPAttnProcessor2_0(c, cd, scale=scale, num_tokens=t, skip=skip), 17, prefix="proc.")
            y = p(attn, hs, encoder_hidden_states=ehs)
            assert rel_rms(y, g[f"ip_skip{int(skip)}"]) < TOL
            if not skip:
                assert rel_rms(p.attn_map, g["attn_map"]) < TOL
        # ControlNet processor is the skip=True path (attention_processor.py:469-621)
        assert rel_rms(g["cn"], g["ip_skip1"]) < 1e-7
        y = om.AttnProcessor2_0()(make_attn(case, cross=False), hs)
        assert rel_rms(y, g["self"]) < TOL


@pytest.mark.parametrize("case", ["cfg2_c1280_L1024_t4", "cfg2_c640_L4096_t4"])
def test_attn_processors_match_reference_golden_at_cfg_shapes(case):
    """a1-a3 at the shapes of the benchmarked forward: the fixture holds sampled rows (fp16) and the row /
    column sums of the verbatim reference classes' outputs"""
    from conftest import cmp_cfg_golden
    from oracle.gen_golden import ATTN_CFG_CASES
    g = torch.load(os.path.join(GOLDEN, f"attn_{case}.pt"))
    b, l, c, h, cd, nt, t, scale = ATTN_CFG_CASES[case]
    hs, ehs = attn_inputs(case)
    attn = make_attn(case, cross=True)
    with torch.no_grad():
        for skip in (False, True):
            p = det_fill(om.IPAttnProcessor2_0(c, cd, scale=scale, num_tokens=t, skip=skip), 17, prefix="proc.")
            cmp_cfg_golden(p(attn, hs, encoder_hidden_states=ehs), g, case, f"ip_skip{int(skip)}", 6e-4)   # fp16 storage of the rows
        cmp_cfg_golden(om.AttnProcessor2_0()(make_attn(case, cross=False), hs), g, case, "self", 6e-4)


def test_ip_scale_zero_equals_skip():
    case = "c128_t32"
    b, l, c, h, cd, nt, t, scale = ATTN_CASES[case]
    hs, ehs = attn_inputs(case)
    attn = make_attn(case, cross=True)
    with torch.no_grad():
        p0 = det_fill(om.IPAttnProcessor2_0(c, cd, scale=0.0, num_tokens=t), 17, prefix="proc.")
        p1 = det_fill(om.IPAttnProcessor2_0(c, cd, num_tokens=t, skip=True), 17, prefix="proc.")
        assert rel_rms(p0(attn, hs, encoder_hidden_states=ehs), p1(attn, hs, encoder_hidden_states=ehs)) < 1e-7


def test_harmony_and_imageproj_match_reference_golden():
    g = torch.load(os.path.join(GOLDEN, "harmony_imageproj.pt"))
    with torch.no_grad():
        ha = det_fill(om.HarmonyAttention(**HA_CFG), 23, prefix="ha.")
        text, img = det_randn((1, 77, 2048), 31), det_randn((1, 1280), 32)
        out = ha(text, img)
        assert rel_rms(out, g["ha_out"]) < TOL
        proj = det_fill(om.ImageProjModel(2048, 1280, 4), 29, prefix="proj.")
        fused = img + out
        assert rel_rms(proj(fused), g["tokens"]) < TOL
        assert rel_rms(proj(torch.zeros_like(fused)), g["uncond_tokens"]) < TOL


def test_mlpproj_matches_reference_golden():
    """MLPProjModel (IPAdapterFull, ip_adapter.py:50-66) restatement vs the reference class's own output"""
    from oracle.gen_golden import MLP_CFG
    g = torch.load(os.path.join(GOLDEN, "mlpproj.pt"))
    with torch.no_grad():
        m = det_fill(om.MLPProjModel(**MLP_CFG), 43, prefix="mlp.")
        y = m(det_randn((1, 64, MLP_CFG["clip_embeddings_dim"]), 47))
    assert y.shape == g["out"].shape and rel_rms(y, g["out"]) < TOL


@pytest.mark.parametrize("name,cfg", [("plusxl", RES_PLUSXL), ("testcfg", RES_TEST)])
def test_resampler_matches_reference_golden(name, cfg):
    g = torch.load(os.path.join(GOLDEN, f"resampler_{name}.pt"))
    with torch.no_grad():
        r = det_fill(om.Resampler(**cfg), 37, prefix="res.")
        y = r(det_randn((g["batch"], 257, cfg["embedding_dim"]), 41))
    # the reference's only test: shape == (B, num_queries + mean_pooled, output_dim) (test_resampler.py:40)
    assert y.shape == (g["batch"], cfg["num_queries"] + cfg.get("num_latents_mean_pooled", 0), cfg["output_dim"])
    assert rel_rms(y, g["out"]) < TOL


@pytest.mark.refshim
@pytest.mark.skipif(not refshim.available(), reason="/root/reference not present")
def test_oracle_state_dicts_interchange_with_reference_live():
    """Same parameter names/shapes as the verbatim reference modules, and identical outputs."""
    ref = refshim.load()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        pairs = [
            (ref.HarmonyAttention(**HA_CFG), om.HarmonyAttention(**HA_CFG)),
            (ref.ImageProjModel(2048, 1280, 4), om.ImageProjModel(2048, 1280, 4)),
            (ref.Resampler(**RES_TEST), om.Resampler(**RES_TEST)),
            (ref.IPAttnProcessor2_0(640, 2048), om.IPAttnProcessor2_0(640, 2048)),
        ]
        for r, o in pairs:
            rs, os_ = r.state_dict(), o.state_dict()
            assert list(rs.keys()) == list(os_.keys())
            assert all(rs[k].shape == os_[k].shape for k in rs)
            o.load_state_dict(rs, strict=True)
        text, img = det_randn((1, 77, 2048), 1), det_randn((1, 1280), 2)
        assert rel_rms(pairs[0][1](text, img), pairs[0][0](text, img)) < TOL
        x = det_randn((2, 257, 1280), 3)
        assert rel_rms(pairs[2][1](x), pairs[2][0](x)) < TOL


@pytest.mark.skipif(not refshim.available(), reason="/root/reference not present")
@pytest.mark.parametrize("case", ["c1280_t4", "c128_t32"])
def test_reference_legacy_processors_equal_the_2_0_golden(case):
    """the reference's torch<2 processors (attention_processor.py:60-241, baddbmm + softmax + bmm) against the golden
    outputs of its 2_0 processors: same math -- which is why the legacy names alias the 2_0 classes here, as
    ip_adapter.py:13-24 does upstream on torch >= 2.  Their attn_map is the true probability tensor (:221-222)."""
    ref = refshim.load()
    g = torch.load(os.path.join(GOLDEN, f"attn_{case}.pt"))
    b, l, c, h, cd, nt, t, scale = ATTN_CASES[case]
    hs, ehs = attn_inputs(case)
    with torch.no_grad():
        attn = make_attn(case, cross=True)
        p = det_fill(ref.IPAttnProcessor(c, cd, scale=scale, num_tokens=t), 17, prefix="proc.")
        y = p(attn, hs, encoder_hidden_states=ehs)
        assert rel_rms(y, g["ip_skip0"]) < 1e-5
        assert p.attn_map.shape[-1] == t and torch.allclose(p.attn_map.sum(-1), torch.ones_like(p.attn_map.sum(-1)), atol=1e-5)
        assert rel_rms(ref.AttnProcessor()(make_attn(case, cross=False), hs), g["self"]) < 1e-5
