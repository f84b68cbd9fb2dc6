"""CPU, build container only: the host orchestration of ``IPAdapterXL`` (rows a8-a10 of SURVEY.md section 8) against
the reference's OWN ``IPAdapterXL`` class (ip_adapter/ip_adapter.py:249-340, imported verbatim through
oracle/refshim.py): which processors ``set_ip_adapter`` installs where, ``get_image_embeds`` (HarmonyAttention fused
into the CLIP embedding, zero-image uncond branch), and what ``generate`` hands to the pipeline (tiling, text || image
token order, pooled embeddings, seed -> generator, kwargs pass-through, scale).  Both adapters get the same fake
pipeline / CLIP objects and the same seeded projection modules (the oracle's, which are pinned to the reference's);
nothing here needs a GPU because the HIP modules are swapped for those CPU modules before any compute."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import modules as om
from oracle import refshim
from oracle.detfill import det_fill, det_randn
from oracle.sdxl_unet import UNet2DConditionModel as OracleUNet
from oracle.sdxl_unet import tiny_config

pytestmark = pytest.mark.skipif(not refshim.available(), reason="reference tree not present")
HALF = torch.float16


class _Pipe:
    def __init__(self, unet):
        self.unet, self.calls, self.encoded = unet, [], []

    def to(self, device):
        return self

    def encode_prompt(self, prompt, num_images_per_prompt=1, do_classifier_free_guidance=True, negative_prompt=None, **kw):
        self.encoded.append((tuple(prompt) if isinstance(prompt, list) else prompt, num_images_per_prompt,
                             tuple(negative_prompt) if isinstance(negative_prompt, list) else negative_prompt))
        n = (len(prompt) if isinstance(prompt, list) else 1) * num_images_per_prompt
        seed = sum(map(ord, "".join(prompt) if isinstance(prompt, list) else prompt)) % 997
        cd = self.unet.config.cross_attention_dim
        return (det_randn((n, 77, cd), seed).to(HALF), det_randn((n, 77, cd), seed + 1).to(HALF),
                det_randn((n, 32), seed + 2).to(HALF), det_randn((n, 32), seed + 3).to(HALF))

    def __call__(self, **kw):
        self.calls.append(kw)
        return type("O", (), {"images": ["img"] * kw["prompt_embeds"].shape[0]})()


class _Proc:                      # CLIPImageProcessor stand-in
    def __call__(self, images=None, return_tensors=None):
        arr = np.stack([np.asarray(im, dtype=np.float32).mean(axis=(0, 1)) for im in images])
        return type("B", (), {"pixel_values": torch.from_numpy(arr)})()


class _Clip(torch.nn.Module):     # CLIPVisionModelWithProjection stand-in: pixel statistics -> [B, 128]
    config = type("C", (), {"projection_dim": 128, "hidden_size": 64})()

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(det_randn((3, 128), 77), requires_grad=False)

    def forward(self, px, output_hidden_states=False):
        return type("O", (), {"image_embeds": ((px.float() / 255.0) @ self.w.float()).to(px.dtype)})()   # an fp16 model returns fp16


def _modules(mod_ns):
    """the projection + HarmonyAttention modules with identical seeded weights, as classes of `mod_ns`"""
    proj = det_fill(mod_ns.ImageProjModel(cross_attention_dim=256, clip_embeddings_dim=128, clip_extra_context_tokens=4), 5)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        ha = mod_ns.HarmonyAttention(image_hidden_size=128, text_context_dim=256, inter_dim=512, cross_heads=8,
                                     reshape_blocks=8, cross_value_dim=64, scale=1.0, fusion_method="cross_attention")
    return proj.to(HALF), det_fill(ha, 3).to(HALF)


def test_ipadapterxl_orchestration_matches_reference():
    import contextlib, io
    ref = refshim.load()
    img = Image.fromarray((np.random.RandomState(0).rand(24, 20, 3) * 255).astype("uint8"))
    kwargs = dict(prompt="lions", negative_prompt="blurry", extra_text="eight sheep", scale=0.6, num_samples=1, seed=42,
                  num_inference_steps=7, guidance_scale=5.0, height=256, width=256)

    # ---- the reference adapter (constructed without __init__: it would download CLIP weights) ----
    r_unet = det_fill(OracleUNet(tiny_config()), 5)
    r = ref.IPAdapterXL.__new__(ref.IPAdapterXL)
    r.device, r.num_tokens, r.pipe = "cpu", 4, _Pipe(r_unet)
    r.image_encoder, r.clip_image_processor = _Clip(), _Proc()
    r.image_proj_model, r.number_class_crossattention = _modules(ref)
    r.set_ip_adapter()                                                            # ip_adapter.py:99-125, verbatim
    with contextlib.redirect_stdout(io.StringIO()):                               # HarmonyAttention prints upstream
        out_r = r.generate(img, number_class_crossattention=r.number_class_crossattention, **kwargs)

    # ---- ours, same fake objects; HIP modules swapped for the (pinned) oracle modules before any compute ----
    from imagharmony_amd.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
    from imagharmony_amd.ip_adapter import IPAdapterXL
    m_unet = det_fill(OracleUNet(tiny_config()), 5)
    m = IPAdapterXL(_Pipe(m_unet), None, None, "cpu", num_tokens=4, inference=True, dtype=HALF,
                    image_encoder=_Clip(), clip_image_processor=_Proc())
    assert m.clip_embeddings_dim == 128
    m.image_proj_model, m.number_class_crossattention = _modules(om)
    out_m = m.generate(pil_image=img, number_class_crossattention=m.number_class_crossattention, **kwargs)

    # set_ip_adapter: same processor kinds / shapes / skip flags under the same names
    rp, mp = r_unet.attn_processors, m_unet.attn_processors
    assert list(rp) == list(mp) and len(rp) > 0
    for k in rp:
        if type(rp[k]).__name__ == "AttnProcessor2_0":      # (the adapter module imports its own copy of the class)
            assert isinstance(mp[k], AttnProcessor2_0), k
        else:
            assert type(rp[k]).__name__ == "IPAttnProcessor2_0" and isinstance(mp[k], IPAttnProcessor2_0), k
            assert (mp[k].hidden_size, mp[k].cross_attention_dim, mp[k].num_tokens, mp[k].skip) == \
                   (rp[k].hidden_size, rp[k].cross_attention_dim, rp[k].num_tokens, rp[k].skip), k
            assert mp[k].to_k_ip.weight.shape == rp[k].to_k_ip.weight.shape
            assert mp[k].scale == rp[k].scale == 0.6                              # generate() -> set_scale
    # generate: identical encode_prompt calls and identical pipeline arguments
    assert r.pipe.encoded == m.pipe.encoded
    kr, km = r.pipe.calls[-1], m.pipe.calls[-1]
    assert out_r == out_m == ["img"]
    assert set(kr) == set(km)
    for k in kr:
        if torch.is_tensor(kr[k]):
            assert kr[k].shape == km[k].shape and torch.allclose(kr[k].float(), km[k].float(), atol=2e-3, rtol=2e-3), k
        elif k == "number_class_crossattention":                            # test.py:38's stray kwarg travels through both
            assert kr[k] is r.number_class_crossattention and km[k] is m.number_class_crossattention
        elif isinstance(kr[k], torch.Generator):
            assert kr[k].initial_seed() == km[k].initial_seed() == 42
        else:
            assert kr[k] == km[k], k
    assert kr["prompt_embeds"].shape == (1, 77 + 4, 256)


class _ClipH(torch.nn.Module):    # vision model with hidden states: the PlusXL path reads hidden_states[-2]
    config = type("C", (), {"projection_dim": 128, "hidden_size": 64})()

    def forward(self, px, output_hidden_states=False):
        base = (px.float() / 255.0).unsqueeze(1).repeat(1, 9, 1)                   # [B, 9 tokens, 3]
        w = det_randn((3, 64), 78)
        hs = [(base * (i + 1)) @ w for i in range(3)]
        return type("O", (), {"hidden_states": [h.to(px.dtype) for h in hs]})()


def test_ipadapterplusxl_orchestration_matches_reference():
    """IPAdapterPlusXL (ip_adapter.py:389-478): Resampler over the penultimate hidden states of the image and of a
    zero image, then the same hand-over to the pipeline; multi-sample tiling (num_samples = 2)"""
    ref = refshim.load()
    img = Image.fromarray((np.random.RandomState(1).rand(24, 20, 3) * 255).astype("uint8"))
    cfg = dict(dim=128, depth=2, dim_head=32, heads=4, num_queries=16, embedding_dim=64, output_dim=256, ff_mult=2)
    kwargs = dict(prompt=None, negative_prompt=None, scale=0.9, num_samples=2, seed=7, num_inference_steps=5, guidance_scale=4.0)

    r = ref.IPAdapterPlusXL.__new__(ref.IPAdapterPlusXL)
    r.device, r.num_tokens, r.pipe = "cpu", 16, _Pipe(det_fill(OracleUNet(tiny_config()), 5))
    r.image_encoder, r.clip_image_processor = _ClipH(), _Proc()
    r.image_proj_model = det_fill(ref.Resampler(**cfg), 9).to(HALF)
    r.set_ip_adapter()
    out_r = r.generate(img, **kwargs)

    from imagharmony_amd.ip_adapter import IPAdapterPlusXL
    m = IPAdapterPlusXL(_Pipe(det_fill(OracleUNet(tiny_config()), 5)), None, None, "cpu", num_tokens=16, dtype=HALF,
                        image_encoder=_ClipH(), clip_image_processor=_Proc())
    assert m.clip_hidden_size == 64
    m.image_proj_model = det_fill(om.Resampler(**cfg), 9).to(HALF)
    out_m = m.generate(pil_image=img, **kwargs)

    assert out_r == out_m == ["img", "img"] and r.pipe.encoded == m.pipe.encoded
    assert r.pipe.encoded[0][0] == ("best quality, high quality",)                 # default prompts (ip_adapter.py:433-436)
    kr, km = r.pipe.calls[-1], m.pipe.calls[-1]
    assert set(kr) == set(km)
    for k in kr:
        if torch.is_tensor(kr[k]):
            assert kr[k].shape == km[k].shape and torch.allclose(kr[k].float(), km[k].float(), atol=3e-3, rtol=3e-3), k
        elif isinstance(kr[k], torch.Generator):
            assert kr[k].initial_seed() == km[k].initial_seed() == 7
        else:
            assert kr[k] == km[k], k
    assert kr["prompt_embeds"].shape == (2, 77 + 16, 256)
    assert all(p.scale == 0.9 for p in m.pipe.unet.attn_processors.values() if hasattr(p, "to_k_ip"))


def test_get_generator_and_checkpoint_loading_match_reference(tmp_path):
    """get_generator (ip_adapter/utils.py:83-93) and load_ip_adapter of an ip_adapter.bin (ip_adapter.py:135-154): both
    adapters end up with the same weights in the projection, the HarmonyAttention module and the 70 -> (here 22)
    IP processors, keyed '<idx>.to_k_ip.weight' in attn_processors order"""
    import contextlib, io
    ref = refshim.load()
    from imagharmony_amd.utils import get_generator
    for seed in (None, 5, [1, 2, 3]):
        a, b = ref.get_generator(seed, "cpu"), get_generator(seed, "cpu")
        if seed is None:
            assert a is None and b is None
        elif isinstance(seed, list):
            assert [g.initial_seed() for g in a] == [g.initial_seed() for g in b] == seed
        else:
            assert a.initial_seed() == b.initial_seed() == seed and torch.equal(torch.randn(4, generator=a), torch.randn(4, generator=b))

    # a checkpoint in the reference's own format, written from a donor reference adapter
    donor_unet = det_fill(OracleUNet(tiny_config()), 5)
    donor = ref.IPAdapterXL.__new__(ref.IPAdapterXL)
    donor.device, donor.num_tokens, donor.pipe = "cpu", 4, _Pipe(donor_unet)
    donor.set_ip_adapter()
    for n, p in donor_unet.attn_processors.items():
        det_fill(p, 21, prefix=n)
    proj, ha = _modules(ref)
    layers = torch.nn.ModuleList(donor_unet.attn_processors.values())
    path = str(tmp_path / "ip_adapter.bin")
    torch.save({"image_proj": det_fill(proj, 31).state_dict(), "ip_adapter": layers.state_dict(),
                "composed_adapter": det_fill(ha, 33).state_dict()}, path)
    assert sorted(layers.state_dict())[0].endswith(".to_k_ip.weight") and len(layers.state_dict()) == 2 * sum(
        hasattr(p, "to_k_ip") for p in donor_unet.attn_processors.values())

    r = ref.IPAdapterXL.__new__(ref.IPAdapterXL)
    r.device, r.num_tokens, r.pipe, r.ip_ckpt = "cpu", 4, _Pipe(det_fill(OracleUNet(tiny_config()), 5)), path
    r.image_proj_model, r.number_class_crossattention = _modules(ref)
    r.set_ip_adapter()
    with contextlib.redirect_stdout(io.StringIO()):
        r.load_ip_adapter()

    from imagharmony_amd.ip_adapter import IPAdapterXL
    from imagharmony_amd.modules import HarmonyAttention
    hm = HarmonyAttention(image_hidden_size=128, text_context_dim=256, inter_dim=512, cross_heads=8, reshape_blocks=8,
                          cross_value_dim=64, scale=1.0, fusion_method="cross_attention")
    m = IPAdapterXL(_Pipe(det_fill(OracleUNet(tiny_config()), 5)), None, path, "cpu", num_tokens=4, inference=True,
                    number_class_crossattention=hm, dtype=HALF, clip_embeddings_dim=128)      # loads in __init__ (ip_adapter.py:89)
    for (kr_, vr), (km_, vm) in zip(r.image_proj_model.state_dict().items(), m.image_proj_model.state_dict().items()):
        assert kr_ == km_ and torch.equal(vr.float(), vm.float())
    for (kr_, vr), (km_, vm) in zip(r.number_class_crossattention.state_dict().items(), m.number_class_crossattention.state_dict().items()):
        assert kr_ == km_ and torch.equal(vr.float(), vm.float())
    rl = torch.nn.ModuleList(r.pipe.unet.attn_processors.values()).state_dict()
    ml = torch.nn.ModuleList(m.pipe.unet.attn_processors.values()).state_dict()
    assert list(rl) == list(ml) and all(torch.equal(rl[k].float(), ml[k].float()) for k in rl)
