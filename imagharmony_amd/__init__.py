"""imagharmony_amd -- the MI355X-native (gfx950) SDXL denoising hot path of IMAGHarmony.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); the
arithmetic runs in hand-written HIP kernels behind the C ABI of ``libimh_hip.so`` (include/imh.h).
"""
__version__ = "0.1.0"

# the reference package's exports (ip_adapter/__init__.py:1-11), resolved lazily so that importing the package
# does not import torch
__all__ = ["IPAdapter", "IPAdapterPlus", "IPAdapterPlusXL", "IPAdapterXL", "IPAdapterFull"]


def __getattr__(name):
    if name in __all__:
        from . import ip_adapter
        return getattr(ip_adapter, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
