"""Deterministic, construction-order-independent parameter / input generation.

TEST INFRASTRUCTURE ONLY.  Both the verbatim reference modules (in the build container)
and the oracle / HIP modules (anywhere) are filled through these helpers, so a golden
output only needs (seed, shapes) to be reproduced -- no weights are committed.
"""
import zlib

import torch


def det_randn(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(int(seed) & 0x7FFFFFFF)
    return (torch.randn(tuple(shape), generator=g, dtype=torch.float32) * scale).to(dtype)


@torch.no_grad()
def det_fill(module, seed, prefix=""):
    """Fill every parameter of ``module`` from (seed, parameter name, shape) only.
    matrices / conv kernels: N(0, 1/fan_in); biases: N(0, 0.05^2); norm weights: 1 + N(0, 0.1^2);
    everything else 1-D or 3-D (e.g. Resampler.latents): N(0, 1/last_dim)."""
    for name, p in module.named_parameters():
        full = prefix + name
        s = (zlib.crc32(full.encode()) ^ (int(seed) * 2654435761)) & 0x7FFFFFFF
        if p.ndim >= 2 and name.endswith("weight"):
            fan_in = p[0].numel()
            v = det_randn(p.shape, s, fan_in ** -0.5)
        elif name.endswith("bias"):
            v = det_randn(p.shape, s, 0.05)
        elif p.ndim == 1 and name.endswith("weight"):
            v = 1.0 + det_randn(p.shape, s, 0.1)
        else:
            v = det_randn(p.shape, s, p.shape[-1] ** -0.5)
        p.copy_(v.to(p.dtype))
    return module
