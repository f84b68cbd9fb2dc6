"""imagharmony_amd -- the MI355X-native (gfx950) SDXL denoising hot path of IMAGHarmony.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); the
arithmetic runs in hand-written HIP kernels behind the C ABI of ``libimh_hip.so`` (include/imh.h).
"""
__version__ = "0.1.0"
