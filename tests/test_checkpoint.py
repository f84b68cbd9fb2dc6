"""CPU: the on-disk formats either side of the path: accelerate checkpoint -> ip_adapter.bin
regrouping (convert_bin.py) checked against the reference's OWN converter when /root/reference is present, and the
.bin / .safetensors readers that feed IPAdapter.load_ip_adapter."""
import importlib.util
import os

import pytest
import torch

from imagharmony_amd import checkpoint as ck


def _fake_training_checkpoint():
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    return {"unet.conv_in.weight": r(8, 4, 3, 3),
            "image_proj_model.proj.weight": r(16, 8), "image_proj_model.proj.bias": r(16),
            "image_proj_model.norm.weight": r(4), "image_proj_model.norm.bias": r(4),
            "adapter_modules.1.to_k_ip.weight": r(8, 6), "adapter_modules.1.to_v_ip.weight": r(8, 6),
            "adapter_modules.3.to_k_ip.weight": r(8, 6), "adapter_modules.3.to_v_ip.weight": r(8, 6),
            "composed_modules.fc1.weight": r(5, 3), "composed_modules.ln.bias": r(5)}


def test_regroup_and_roundtrip_both_formats(tmp_path):
    sd = _fake_training_checkpoint()
    src = tmp_path / "pytorch_model.bin"
    torch.save(sd, src)
    for ext in (".bin", ".safetensors"):
        dst = str(tmp_path / ("ip_adapter" + ext))
        assert ck.convert_checkpoint_to_ip_adapter(str(src), dst)
        got = ck.load_ip_adapter_file(dst)
        assert sorted(got) == ["composed_adapter", "image_proj", "ip_adapter"]
        assert list(got["ip_adapter"]) == ["1.to_k_ip.weight", "1.to_v_ip.weight", "3.to_k_ip.weight", "3.to_v_ip.weight"] \
            or sorted(got["ip_adapter"]) == ["1.to_k_ip.weight", "1.to_v_ip.weight", "3.to_k_ip.weight", "3.to_v_ip.weight"]
        assert torch.equal(got["image_proj"]["proj.weight"], sd["image_proj_model.proj.weight"])
        assert torch.equal(got["composed_adapter"]["ln.bias"], sd["composed_modules.ln.bias"])
        assert not any(k.startswith("unet") for grp in got.values() for k in grp)
    assert not ck.convert_checkpoint_to_ip_adapter(str(tmp_path / "missing.bin"), str(tmp_path / "x.bin"))
    torch.save({"unet.a": torch.zeros(1)}, tmp_path / "only_unet.bin")
    assert not ck.convert_checkpoint_to_ip_adapter(str(tmp_path / "only_unet.bin"), str(tmp_path / "y.bin"))
    assert not os.path.exists(tmp_path / "y.bin")


def test_convert_tree_layout(tmp_path):
    for run, step in (("run_a", 100), ("run_a", 200), ("run_b", 100)):
        d = tmp_path / run / f"checkpoint-{step}"
        d.mkdir(parents=True)
        torch.save(_fake_training_checkpoint(), d / "pytorch_model.bin")
    (tmp_path / "run_b" / "checkpoint-300").mkdir()                 # no source file -> skipped
    assert ck.convert_tree(str(tmp_path)) == (3, 1, 0)
    assert ck.convert_tree(str(tmp_path)) == (0, 4, 0)              # outputs exist -> all skipped
    assert os.path.exists(tmp_path / "run_a" / "checkpoint-200" / "ip_adapter.bin")


@pytest.mark.skipif(not os.path.exists("/root/reference/convert_bin.py"), reason="reference tree not present")
def test_matches_reference_converter(tmp_path, capsys):
    spec = importlib.util.spec_from_file_location("_ref_convert_bin", "/root/reference/convert_bin.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    src = tmp_path / "pytorch_model.bin"
    torch.save(_fake_training_checkpoint(), src)
    assert ref.convert_checkpoint_to_ip_adapter(str(src), str(tmp_path / "ref.bin"))
    assert ck.convert_checkpoint_to_ip_adapter(str(src), str(tmp_path / "ours.bin"))
    a, b = torch.load(tmp_path / "ref.bin"), torch.load(tmp_path / "ours.bin")
    assert list(a) == list(b)
    for grp in a:
        assert list(a[grp]) == list(b[grp])
        for k in a[grp]:
            assert torch.equal(a[grp][k], b[grp][k])
