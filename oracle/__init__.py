"""CPU oracle for the IMAGHarmony SDXL denoising hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product path (``imagharmony_amd``) never routes through
it and fails loudly when the HIP extension is missing.

What it restates (plain PyTorch, fp32 on CPU), with the reference file:line each
piece follows:

* ``oracle.modules``    -- IPAttnProcessor2_0 / AttnProcessor2_0
                           (ip_adapter/attention_processor.py:244-465),
                           Cross_Attention (:12-56), HarmonyAttention
                           (train.py:188-266), ImageProjModel
                           (ip_adapter/ip_adapter.py:28-48), Resampler
                           (ip_adapter/resampler.py:13-158).
* ``oracle.sdxl_unet``  -- diffusers==0.30.0 ``UNet2DConditionModel`` (SDXL
                           config) + ``Attention`` with the processor protocol.
                           diffusers is a third-party dependency pinned in the
                           reference's requirements.txt:25 and is NOT vendored
                           in /root/reference nor installed here, so this is a
                           restatement of its published architecture
                           (SURVEY.md Appendix A); validated by the exact SDXL
                           parameter count 2,567,463,684 and the state-dict key
                           schema.
* ``oracle.schedulers`` -- DDIM (eta=0) / EulerDiscrete.
* ``oracle.pipeline``   -- the denoise loop of
                           ip_adapter/custom_pipelines.py:249-363.

Pinning status
--------------
The reference ships no golden vectors, known-answer tests or fixtures for this
path (its only test asserts a tensor shape, ip_adapter/test_resampler.py:40).
``oracle.modules`` is pinned against the reference's OWN python modules imported
verbatim from /root/reference (``oracle/refshim.py``); the generated vectors are
committed under ``tests/golden/`` together with ``oracle/gen_golden.py``.
``oracle.pipeline`` (the denoise loop) is pinned against the reference's OWN
``StableDiffusionXLCustomPipeline.__call__`` executed verbatim on a restated base
class (``refshim.load_pipeline_class``, tests/test_oracle_loop_vs_reference.py).
``oracle.sdxl_unet``, ``oracle.schedulers``, ``oracle.vae`` and
``oracle.pipeline.rescale_noise_cfg`` restate diffusers, which cannot be executed
here: for those pieces **parity is unpinned** beyond the parameter counts
(UNet 2,567,463,684; VAE 83,653,863) / key-schema / closed-form checks (stated
again in DESIGN.md).
"""
