"""On-disk formats either side of the path: the accelerate training checkpoint -> ``ip_adapter.bin``
regrouping of the reference's convert_bin.py, and readers for the grouped file (``.bin`` via torch.load, or
``.safetensors`` with ``<group>.<key>`` names).  Pure host code: no arithmetic, nothing here touches the GPU.

ip_adapter.bin = {"image_proj": {...}, "ip_adapter": {"<idx>.to_k_ip.weight", "<idx>.to_v_ip.weight", ...},
"composed_adapter": {HarmonyAttention keys}} -- what ``IPAdapter.load_ip_adapter`` consumes (ip_adapter.py:135-154)."""
import os
from collections import OrderedDict
from typing import Dict

import torch

# training-checkpoint prefix -> group of ip_adapter.bin (convert_bin.py:21-32)
PREFIXES = (("image_proj_model.", "image_proj"), ("adapter_modules.", "ip_adapter"), ("composed_modules.", "composed_adapter"))


def regroup(state_dict: Dict[str, torch.Tensor]) -> Dict[str, "OrderedDict[str, torch.Tensor]"]:
    """flat accelerate state dict (``unet.`` / ``image_proj_model.`` / ``adapter_modules.`` / ``composed_modules.``
    prefixes) -> the three groups; UNet weights are dropped (they stay frozen)."""
    out = {g: OrderedDict() for _, g in PREFIXES}
    for k, v in state_dict.items():
        for pre, g in PREFIXES:
            if k.startswith(pre):
                out[g][k[len(pre):]] = v
                break
    return out


def convert_checkpoint_to_ip_adapter(pytorch_model_path, output_ip_adapter_path) -> bool:
    """same contract as convert_bin.py:5-48: False (nothing written) if the source is missing or holds none of the
    expected prefixes; output format chosen by extension (.bin -> torch.save, .safetensors -> flat '<group>.<key>')"""
    if not os.path.exists(pytorch_model_path):
        return False
    sd = torch.load(pytorch_model_path, map_location="cpu")
    groups = regroup(sd)
    if not any(groups.values()):
        return False
    save_ip_adapter(groups, output_ip_adapter_path)
    return True


def save_ip_adapter(groups, path):
    if os.path.splitext(path)[-1] == ".safetensors":
        from safetensors.torch import save_file
        save_file({f"{g}.{k}": v.contiguous() for g, d in groups.items() for k, v in d.items()}, path)
    else:
        torch.save({g: dict(d) for g, d in groups.items()}, path)


def load_ip_adapter_file(path):
    """-> {"image_proj": {...}, "ip_adapter": {...}, "composed_adapter": {...}} from either format
    (the reference's own .safetensors branch, ip_adapter.py:137-147, raises KeyError on 'composed_adapter')"""
    if os.path.splitext(path)[-1] == ".safetensors":
        from safetensors import safe_open
        out = {g: {} for _, g in PREFIXES}
        with safe_open(path, framework="pt", device="cpu") as f:
            for key in f.keys():
                for g in out:
                    if key.startswith(g + "."):
                        out[g][key[len(g) + 1:]] = f.get_tensor(key)
        return out
    sd = torch.load(path, map_location="cpu")
    sd.setdefault("composed_adapter", {})
    return sd


def convert_tree(base_log_dir):
    """convert_bin.py:58-118: every <run>/checkpoint-*/pytorch_model.bin -> ip_adapter.bin next to it.
    Returns (converted, skipped, errors)."""
    conv = skip = err = 0
    for run in sorted(os.listdir(base_log_dir)):
        rd = os.path.join(base_log_dir, run)
        if not os.path.isdir(rd):
            continue
        for ck in sorted(os.listdir(rd)):
            cd = os.path.join(rd, ck)
            if not (ck.startswith("checkpoint-") and os.path.isdir(cd)):
                continue
            src, dst = os.path.join(cd, "pytorch_model.bin"), os.path.join(cd, "ip_adapter.bin")
            if os.path.exists(dst) or not os.path.exists(src):
                skip += 1
                continue
            try:
                if convert_checkpoint_to_ip_adapter(src, dst):
                    conv += 1
                else:
                    err += 1
            except Exception:      # noqa: BLE001 -- same accounting as the reference: count and continue
                err += 1
    return conv, skip, err


if __name__ == "__main__":
    import sys
    print("converted %d, skipped %d, errors %d" % convert_tree(sys.argv[1]))
