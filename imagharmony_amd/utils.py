"""ip_adapter/utils.py:80-93."""
import torch
import torch.nn.functional as F


def is_torch2_available():
    return hasattr(F, "scaled_dot_product_attention")


def get_generator(seed, device):
    """int -> one generator; list of ints -> one generator per sample (the PNS seed fan-out hook).
    The reference seeds a device generator, which is not reproducible across backends (SURVEY.md
    Appendix D.9); pass device='cpu' for backend-independent noise (the default used by this package)."""
    if seed is None:
        return None
    if isinstance(seed, list):
        return [torch.Generator(device).manual_seed(s) for s in seed]
    return torch.Generator(device).manual_seed(seed)
