"""CPU: the SDXL encode_prompt glue (imagharmony_amd.text) with tiny random transformers CLIP text models and a stub
tokenizer (no vocabulary files offline): shapes, concatenation order, pooled source, negative-prompt rules, tiling."""
import pytest
import torch
from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection

from imagharmony_amd.text import SDXLPromptEncoder


class _Tok:
    model_max_length = 12

    def __call__(self, prompts, padding=None, max_length=None, truncation=None, return_tensors=None):
        assert padding == "max_length" and max_length == self.model_max_length and truncation and return_tensors == "pt"
        ids = torch.zeros(len(prompts), max_length, dtype=torch.long)
        for i, p in enumerate(prompts):
            toks = [1] + [3 + (ord(c) % 90) for c in p][:max_length - 2] + [2]
            ids[i, :len(toks)] = torch.tensor(toks)
        return type("B", (), {"input_ids": ids})()


def _models():
    torch.manual_seed(0)
    c1 = CLIPTextConfig(vocab_size=100, hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                        max_position_embeddings=12, projection_dim=32)
    c2 = CLIPTextConfig(vocab_size=100, hidden_size=48, intermediate_size=96, num_hidden_layers=3, num_attention_heads=4,
                        max_position_embeddings=12, projection_dim=40)
    return CLIPTextModel(c1).eval(), CLIPTextModelWithProjection(c2).eval()


def test_encode_prompt_semantics():
    e1, e2 = _models()
    enc = SDXLPromptEncoder(_Tok(), _Tok(), e1, e2)
    pe, ne, pp, npp = enc("eight sheep", num_images_per_prompt=2, negative_prompt=None)
    assert pe.shape == (2, 12, 32 + 48) and pp.shape == (2, 40)                  # hidden dims concatenated; pooled = projection of encoder 2
    assert torch.equal(pe[0], pe[1]) and torch.count_nonzero(ne) == 0 and torch.count_nonzero(npp) == 0     # force_zeros_for_empty_prompt
    ids = _Tok()(["eight sheep"], "max_length", 12, True, "pt").input_ids
    with torch.no_grad():
        h1 = e1(ids, output_hidden_states=True).hidden_states[-2]
        o2 = e2(ids, output_hidden_states=True)
    assert torch.allclose(pe[0, :, :32], h1[0]) and torch.allclose(pe[0, :, 32:], o2.hidden_states[-2][0])
    assert torch.allclose(pp[0], o2.text_embeds[0])
    pe2, ne2, _, npp2 = enc(["a", "b"], negative_prompt="low quality")
    assert pe2.shape == (2, 12, 80) and ne2.shape == (2, 12, 80) and torch.count_nonzero(ne2) > 0
    assert torch.equal(ne2[0], ne2[1]) and not torch.equal(pe2[0], pe2[1])
    soft = SDXLPromptEncoder(_Tok(), _Tok(), e1, e2, force_zeros_for_empty_prompt=False)
    _, ne3, _, _ = soft("x")
    assert torch.count_nonzero(ne3) > 0                                            # empty string is encoded instead
    assert enc("x", do_classifier_free_guidance=False)[1] is None
    with pytest.raises(ValueError):
        enc(["a", "b"], negative_prompt=["only one"])


def test_pipeline_uses_the_encoder_object():
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    e1, e2 = _models()
    pipe = StableDiffusionXLCustomPipeline.__new__(StableDiffusionXLCustomPipeline)
    pipe.text_encoder = SDXLPromptEncoder(_Tok(), _Tok(), e1, e2)
    out = pipe.encode_prompt("lions", num_images_per_prompt=1, do_classifier_free_guidance=True, negative_prompt="blurry")
    assert len(out) == 4 and out[0].shape == (1, 12, 80)
