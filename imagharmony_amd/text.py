"""``encode_prompt`` for the SDXL pipeline with stock ``transformers`` text encoders -- the step before the hot path
(reference call sites ip_adapter.py:285-297,308-319 -> diffusers StableDiffusionXLPipeline.encode_prompt,
restated here because diffusers is absent).  No kernels of ours: the two CLIP text models run as ordinary torch modules.

    enc = SDXLPromptEncoder(tokenizer, tokenizer_2, text_encoder, text_encoder_2)
    pipe = StableDiffusionXLCustomPipeline(unet, text_encoder=enc, ...)

Semantics kept from diffusers 0.30.0: every (tokenizer, encoder) pair tokenises to ``model_max_length`` with truncation,
the penultimate hidden state of each encoder is concatenated on the feature axis (768 + 1280 = 2048), the pooled
embedding is the projected output of the LAST encoder, an absent negative prompt becomes zeros when
``force_zeros_for_empty_prompt`` (SDXL-base default) and the empty string otherwise, results are tiled
``num_images_per_prompt`` times.
"""
from typing import List, Optional, Union

import torch


class SDXLPromptEncoder:
    def __init__(self, tokenizer, tokenizer_2, text_encoder, text_encoder_2, force_zeros_for_empty_prompt=True, device=None):
        self.pairs = [(t, e) for t, e in ((tokenizer, text_encoder), (tokenizer_2, text_encoder_2)) if t is not None and e is not None]
        if not self.pairs:
            raise ValueError("at least one (tokenizer, text_encoder) pair is needed")
        self.force_zeros = force_zeros_for_empty_prompt
        self.device = device

    @torch.no_grad()
    def _encode(self, prompts: List[str]):
        embeds, pooled = [], None
        for tok, enc in self.pairs:
            dev = self.device or next(enc.parameters()).device
            ids = tok(prompts, padding="max_length", max_length=tok.model_max_length, truncation=True,
                      return_tensors="pt").input_ids
            out = enc(ids.to(dev), output_hidden_states=True)
            pooled = out[0]                                   # of the last encoder: CLIPTextModelWithProjection.text_embeds
            embeds.append(out.hidden_states[-2])
        return torch.cat(embeds, dim=-1), pooled

    def __call__(self, prompt: Union[str, List[str]], num_images_per_prompt: int = 1, do_classifier_free_guidance: bool = True,
                 negative_prompt: Optional[Union[str, List[str]]] = None, **_):
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        pe, pooled = self._encode(prompts)
        ne = npooled = None
        if do_classifier_free_guidance:
            if negative_prompt is None and self.force_zeros:
                ne, npooled = torch.zeros_like(pe), torch.zeros_like(pooled)
            else:
                neg = negative_prompt if negative_prompt is not None else ""
                negs = [neg] * len(prompts) if isinstance(neg, str) else list(neg)
                if len(negs) != len(prompts):
                    raise ValueError(f"negative_prompt has batch size {len(negs)}, prompt has {len(prompts)}")
                ne, npooled = self._encode(negs)

        def tile(t, seq):
            if t is None:
                return None
            b = t.shape[0]
            return (t.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, t.shape[1], -1) if seq
                    else t.repeat(1, num_images_per_prompt).view(b * num_images_per_prompt, -1))

        return tile(pe, True), tile(ne, True), tile(pooled, False), tile(npooled, False)
