"""Import the reference's OWN python modules verbatim from /root/reference.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: /root/reference
does not exist on the GPU box, so nothing in ``-m gpu`` tests, smoke() or bench.py
may import this file.  It exists to (a) pin ``oracle.modules`` against the real
reference code and (b) mint ``tests/golden/*`` (oracle/gen_golden.py).

diffusers / torchvision are absent and ``tutorial_train_sdxl_ori`` does not exist even
upstream (ip_adapter/ip_adapter.py:10), so they are stubbed in sys.modules before the
reference files are imported.
"""
import contextlib
import importlib.machinery
import importlib.util
import io
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "ip_adapter"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def _load_file(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def load():
    """Returns a namespace with the reference classes: IPAttnProcessor2_0, AttnProcessor2_0,
    IPAttnProcessor (legacy), AttnProcessor (legacy), Cross_Attention, Resampler,
    HarmonyAttention, ImageProjModel."""
    if "ns" in _cache:
        return _cache["ns"]
    if not available():
        raise RuntimeError("/root/reference is not present (refshim only works in the build container)")
    sys.dont_write_bytecode = True
    ap = _load_file("_ref_attention_processor", os.path.join(REF, "ip_adapter", "attention_processor.py"))
    rs = _load_file("_ref_resampler", os.path.join(REF, "ip_adapter", "resampler.py"))
    ut = _load_file("_ref_utils", os.path.join(REF, "ip_adapter", "utils.py"))

    # train.py / ip_adapter/ip_adapter.py need third-party stubs
    import transformers  # noqa: F401  (must be imported before torchvision is stubbed)
    from transformers import CLIPImageProcessor, CLIPVisionModelWithProjection  # noqa: F401

    class _D:
        def __init__(self, *a, **k):
            pass

    saved = {k: sys.modules.get(k) for k in list(sys.modules)}
    _stub("diffusers", StableDiffusionPipeline=_D, StableDiffusionXLPipeline=type("StableDiffusionXLPipeline", (), {}),
          AutoencoderKL=_D, DDPMScheduler=_D, UNet2DConditionModel=_D)
    _stub("diffusers.models")
    _stub("diffusers.models.attention_processor", Attention=_D)
    _stub("diffusers.pipelines")
    _stub("diffusers.pipelines.controlnet", MultiControlNetModel=_D)
    _stub("diffusers.pipelines.stable_diffusion_xl", StableDiffusionXLPipelineOutput=_D)
    _stub("diffusers.pipelines.stable_diffusion_xl.pipeline_stable_diffusion_xl", rescale_noise_cfg=lambda *a, **k: None)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    _stub("tutorial_train_sdxl_ori", HarmonyAttention=None, ComposedAttention=None)
    sys.path.insert(0, REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import train as ref_train
            import ip_adapter.ip_adapter as ref_ipa
    finally:
        sys.path.remove(REF)
        # drop the stubs and the reference's top-level packages again so they cannot leak
        for k in list(sys.modules):
            if k not in saved and not k.startswith("_ref_"):
                if k.split(".")[0] in ("diffusers", "torchvision", "tutorial_train_sdxl_ori", "ip_adapter",
                                       "train", "baseline", "shared_models"):
                    sys.modules.pop(k, None)

    ns = types.SimpleNamespace(
        IPAttnProcessor2_0=ap.IPAttnProcessor2_0, AttnProcessor2_0=ap.AttnProcessor2_0,
        IPAttnProcessor=ap.IPAttnProcessor, AttnProcessor=ap.AttnProcessor,
        CNAttnProcessor2_0=ap.CNAttnProcessor2_0,
        Cross_Attention=ap.Cross_Attention, Resampler=rs.Resampler, get_generator=ut.get_generator,
        HarmonyAttention=ref_train.HarmonyAttention, ImageProjModel=ref_ipa.ImageProjModel,
        MLPProjModel=ref_ipa.MLPProjModel,
        IPAdapterXL=ref_ipa.IPAdapterXL, IPAdapterPlusXL=ref_ipa.IPAdapterPlusXL)
    _cache["ns"] = ns
    return ns


# ---------------------------------------------------------------------------------------------------------------
# The reference's own denoise loop (ip_adapter/custom_pipelines.py:23-389), executed verbatim.
# Its base class, diffusers' StableDiffusionXLPipeline, is absent; ``_PipeBase`` below supplies only the helpers the
# loop calls (restated from diffusers 0.30.0: prepare_latents scales by init_noise_sigma, _get_add_time_ids
# concatenates original_size + crops + target_size, prepare_extra_step_kwargs is empty for schedulers without
# eta/generator, encode_prompt passes pre-computed embeddings through).  UNet and scheduler are injected objects.
class _PipeBase:
    vae_scale_factor = 8
    text_encoder_2 = None

    @property
    def _execution_device(self):
        return "cpu"

    def check_inputs(self, *a, **k):
        pass

    def encode_prompt(self, prompt=None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                      negative_pooled_prompt_embeds=None, **k):
        return prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        assert latents is not None, "the parity tests always pass the initial noise"
        return latents.to(dtype) * self.scheduler.init_noise_sigma

    def prepare_extra_step_kwargs(self, generator, eta):
        return {}

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, dtype, text_encoder_projection_dim=None):
        import torch
        return torch.tensor([list(original_size) + list(crops_coords_top_left) + list(target_size)], dtype=dtype)

    def progress_bar(self, total=None):
        class _PB:
            def __enter__(s):
                return s

            def __exit__(s, *a):
                return False

            def update(s):
                pass
        return _PB()

    def maybe_free_model_hooks(self):
        pass


def load_pipeline_class():
    """-> the reference's StableDiffusionXLCustomPipeline class (its __call__ body runs verbatim) on top of _PipeBase"""
    if "pipe" in _cache:
        return _cache["pipe"]
    if not available():
        raise RuntimeError("/root/reference is not present (refshim only works in the build container)")
    from . import pipeline as opipe

    class _Out:
        def __init__(self, images=None):
            self.images = images

    saved = {k: sys.modules.get(k) for k in list(sys.modules)}
    _stub("diffusers", StableDiffusionXLPipeline=_PipeBase)
    _stub("diffusers.pipelines")
    _stub("diffusers.pipelines.stable_diffusion_xl", StableDiffusionXLPipelineOutput=_Out)
    _stub("diffusers.pipelines.stable_diffusion_xl.pipeline_stable_diffusion_xl", rescale_noise_cfg=opipe.rescale_noise_cfg)
    pkg = types.ModuleType("_ref_ipa_pkg")
    pkg.__path__ = [os.path.join(REF, "ip_adapter")]
    sys.modules["_ref_ipa_pkg"] = pkg
    try:
        _load_file("_ref_ipa_pkg.utils", os.path.join(REF, "ip_adapter", "utils.py"))
        ap = _load_file("_ref_ipa_pkg.attention_processor", os.path.join(REF, "ip_adapter", "attention_processor.py"))
        cp = _load_file("_ref_ipa_pkg.custom_pipelines", os.path.join(REF, "ip_adapter", "custom_pipelines.py"))
    finally:
        for k in list(sys.modules):
            if k not in saved and (k.split(".")[0] == "diffusers"):
                sys.modules.pop(k, None)
    _cache["pipe"] = (cp.StableDiffusionXLCustomPipeline, ap)
    return _cache["pipe"]
