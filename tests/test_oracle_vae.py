"""CPU: structural pins of the restated SDXL VAE (diffusers is absent: parity unpinned) and the
algebra of tiled decoding and post-processing."""
import numpy as np
import torch

from oracle.detfill import det_fill, det_randn
from oracle.vae import AutoencoderKL, postprocess, sdxl_vae_config, tiny_vae_config


def test_sdxl_vae_param_count_and_key_schema():
    with torch.device("meta"):
        m = AutoencoderKL(sdxl_vae_config())
    assert sum(p.numel() for p in m.parameters()) == 83_653_863          # the published size of the SD / SDXL VAE
    assert sum(p.numel() for p in m.decoder.parameters()) == 49_490_179
    sd = m.state_dict()
    for k, shape in [("decoder.conv_in.weight", (512, 4, 3, 3)), ("decoder.mid_block.attentions.0.to_q.weight", (512, 512)),
                     ("decoder.mid_block.attentions.0.group_norm.bias", (512,)),
                     ("decoder.mid_block.attentions.0.to_out.0.bias", (512,)),
                     ("decoder.up_blocks.0.upsamplers.0.conv.weight", (512, 512, 3, 3)),
                     ("decoder.up_blocks.2.resnets.0.conv_shortcut.weight", (256, 512, 1, 1)),
                     ("decoder.up_blocks.3.resnets.2.conv2.weight", (128, 128, 3, 3)),
                     ("decoder.conv_out.weight", (3, 128, 3, 3)), ("post_quant_conv.weight", (4, 4, 1, 1)),
                     ("quant_conv.bias", (8,)), ("encoder.down_blocks.0.downsamplers.0.conv.weight", (128, 128, 3, 3)),
                     ("encoder.conv_out.weight", (8, 512, 3, 3))]:
        assert tuple(sd[k].shape) == shape, k
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd
    assert m.tile_sample_min_size == 512 and m.tile_latent_min_size == 64


def test_sdxl_vae_state_dict_equals_the_published_manifest():
    """Set-equality of (key, shape) between the restated AutoencoderKL, tests/golden/sdxl_vae_manifest.txt (the SDXL VAE's state dict
    enumerated from its published config.json, oracle/gen_manifest.py; 83,653,863 parameters) and -- for the decode half it implements --
    the product's AutoencoderKL."""
    import os
    from conftest import GOLDEN
    from oracle.gen_manifest import numel, read, vae_manifest
    man = read(os.path.join(GOLDEN, "sdxl_vae_manifest.txt"))
    assert man == dict(vae_manifest()) and numel(man.items()) == 83_653_863 and len(man) == 248
    with torch.device("meta"):
        m = AutoencoderKL(sdxl_vae_config())
    osd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert set(osd) == set(man), (sorted(set(osd) - set(man))[:5], sorted(set(man) - set(osd))[:5])
    assert osd == man
    from imagharmony_amd.vae import AutoencoderKL as HV
    with torch.device("meta"):
        h = HV()
    hsd = {k: tuple(v.shape) for k, v in h.state_dict().items()}
    dec = {k: v for k, v in man.items() if k.startswith(("decoder.", "post_quant_conv."))}
    assert set(hsd) == set(dec), (sorted(set(hsd) - set(dec))[:5], sorted(set(dec) - set(hsd))[:5])
    assert hsd == dec


def test_decode_shapes_tiling_and_postprocess():
    vae = det_fill(AutoencoderKL(tiny_vae_config()), 3).eval()
    z = det_randn((1, 4, 48, 40), 5)
    with torch.no_grad():
        full = vae.decode(z)
        assert full.shape == (1, 3, 384, 320)
        assert vae.tile_latent_min_size == 32 and vae.tile_sample_min_size == 256
        vae.enable_tiling()
        tiled = vae.decode(z)
    assert tiled.shape == full.shape and torch.isfinite(tiled).all()
    # a tile's interior is decoded from the same latents: away from tile seams and borders the two agree loosely
    # (GroupNorm statistics differ per tile), and tiling a latent that fits one tile is the identity
    small = det_randn((1, 4, 32, 32), 6)
    with torch.no_grad():
        assert torch.equal(vae.decode(small), vae.decoder(vae.post_quant_conv(small)))
    img = postprocess(full, "np")
    assert img.shape == (1, 384, 320, 3) and img.min() >= 0.0 and img.max() <= 1.0
    pil = postprocess(full, "pil")
    assert pil[0].size == (320, 384) and np.asarray(pil[0]).dtype == np.uint8
    assert torch.equal(postprocess(full, "pt"), (full / 2 + 0.5).clamp(0, 1))


def test_encoder_downsamples_by_eight():
    vae = det_fill(AutoencoderKL(tiny_vae_config()), 3).eval()
    with torch.no_grad():
        moments = vae.quant_conv(vae.encoder(det_randn((1, 3, 64, 64), 7)))
    assert moments.shape == (1, 8, 8, 8)
