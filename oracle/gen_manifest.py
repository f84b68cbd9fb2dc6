"""TEST INFRASTRUCTURE (never imported by the product): writes tests/golden/sdxl_unet_manifest.txt and sdxl_vae_manifest.txt -- one
"<state-dict key> <shape>" line per tensor of stabilityai/stable-diffusion-xl-base-1.0's `unet` and `vae` (diffusers 0.30.0 key schema,
requirements.txt:25 of the reference; the call sites are ip_adapter/custom_pipelines.py:338-345 and :365-377).

diffusers is not installed here and there is no network, so the manifests are NOT downloaded: they are enumerated from the published
`config.json` values below by string templates -- deliberately a different mechanism from oracle/sdxl_unet.py and oracle/vae.py (which
build nn.Module trees), so that a renamed / mis-shaped tensor in either restatement, or in the product's model classes, fails
set-equality (tests/test_oracle_unet.py, tests/test_oracle_vae.py).  The two published totals the enumeration must reproduce are checked
here and in the tests: 2,567,463,684 UNet parameters, 83,653,863 VAE parameters.   Run: python -m oracle.gen_manifest"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")

# unet/config.json (SDXL base 1.0)
UNET = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
            transformer_layers_per_block=(1, 2, 10), cross_attention_dim=2048, addition_time_embed_dim=256,
            projection_class_embeddings_input_dim=2816, time_embed_dim=1280)
# vae/config.json
VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2)


def _lin(out, pfx, o, i, bias=True):
    out.append((pfx + ".weight", (o, i)))
    if bias:
        out.append((pfx + ".bias", (o,)))


def _conv(out, pfx, o, i, k):
    out.append((pfx + ".weight", (o, i, k, k)))
    out.append((pfx + ".bias", (o,)))


def _norm(out, pfx, c):
    out.append((pfx + ".weight", (c,)))
    out.append((pfx + ".bias", (c,)))


def _resnet(out, pfx, cin, cout, temb):
    _norm(out, pfx + ".norm1", cin)
    _conv(out, pfx + ".conv1", cout, cin, 3)
    if temb:
        _lin(out, pfx + ".time_emb_proj", cout, temb)
    _norm(out, pfx + ".norm2", cout)
    _conv(out, pfx + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(out, pfx + ".conv_shortcut", cout, cin, 1)


def _transformer(out, pfx, c, layers, cross):
    _norm(out, pfx + ".norm", c)
    _lin(out, pfx + ".proj_in", c, c)                       # use_linear_projection = true
    for l in range(layers):
        b = f"{pfx}.transformer_blocks.{l}"
        _norm(out, b + ".norm1", c)
        for n in ("to_q", "to_k", "to_v"):
            _lin(out, f"{b}.attn1.{n}", c, c, bias=False)
        _lin(out, b + ".attn1.to_out.0", c, c)
        _norm(out, b + ".norm2", c)
        _lin(out, b + ".attn2.to_q", c, c, bias=False)
        _lin(out, b + ".attn2.to_k", c, cross, bias=False)
        _lin(out, b + ".attn2.to_v", c, cross, bias=False)
        _lin(out, b + ".attn2.to_out.0", c, c)
        _norm(out, b + ".norm3", c)
        _lin(out, b + ".ff.net.0.proj", 8 * c, c)           # GEGLU: 2 x (4 c)
        _lin(out, b + ".ff.net.2", c, 4 * c)
    _lin(out, pfx + ".proj_out", c, c)


def unet_manifest(cfg=UNET):
    out = []
    boc, te, cross = cfg["block_out_channels"], cfg["time_embed_dim"], cfg["cross_attention_dim"]
    _conv(out, "conv_in", boc[0], cfg["in_channels"], 3)
    _lin(out, "time_embedding.linear_1", te, boc[0]); _lin(out, "time_embedding.linear_2", te, te)
    _lin(out, "add_embedding.linear_1", te, cfg["projection_class_embeddings_input_dim"]); _lin(out, "add_embedding.linear_2", te, te)
    ch = boc[0]
    skips = [ch]
    for i, (typ, c) in enumerate(zip(cfg["down_block_types"], boc)):
        for j in range(cfg["layers_per_block"]):
            _resnet(out, f"down_blocks.{i}.resnets.{j}", ch, c, te)
            if typ.startswith("CrossAttn"):
                _transformer(out, f"down_blocks.{i}.attentions.{j}", c, cfg["transformer_layers_per_block"][i], cross)
            ch = c
            skips.append(ch)
        if i < len(boc) - 1:
            _conv(out, f"down_blocks.{i}.downsamplers.0.conv", c, c, 3)
            skips.append(ch)
    _resnet(out, "mid_block.resnets.0", ch, ch, te)
    _transformer(out, "mid_block.attentions.0", ch, cfg["transformer_layers_per_block"][-1], cross)
    _resnet(out, "mid_block.resnets.1", ch, ch, te)
    rboc = boc[::-1]
    rtl = cfg["transformer_layers_per_block"][::-1]
    for i, (typ, c) in enumerate(zip(cfg["up_block_types"], rboc)):
        for j in range(cfg["layers_per_block"] + 1):
            sk = skips.pop()
            _resnet(out, f"up_blocks.{i}.resnets.{j}", ch + sk, c, te)
            if typ.startswith("CrossAttn"):
                _transformer(out, f"up_blocks.{i}.attentions.{j}", c, rtl[i], cross)
            ch = c
        if i < len(boc) - 1:
            _conv(out, f"up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    assert not skips
    _norm(out, "conv_norm_out", boc[0])
    _conv(out, "conv_out", cfg["out_channels"], boc[0], 3)
    return out


def _vae_attn(out, pfx, c):
    _norm(out, pfx + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(out, f"{pfx}.{n}", c, c)


def vae_manifest(cfg=VAE):
    out = []
    boc, L, z = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
    _conv(out, "encoder.conv_in", boc[0], cfg["in_channels"], 3)
    ch = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            _resnet(out, f"encoder.down_blocks.{i}.resnets.{j}", ch, c, 0)
            ch = c
        if i < len(boc) - 1:
            _conv(out, f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
    for side in ("encoder", "decoder"):
        if side == "decoder":
            _conv(out, "decoder.conv_in", boc[-1], z, 3)
            ch = boc[-1]
        _resnet(out, f"{side}.mid_block.resnets.0", ch, ch, 0)
        _vae_attn(out, f"{side}.mid_block.attentions.0", ch)
        _resnet(out, f"{side}.mid_block.resnets.1", ch, ch, 0)
        if side == "encoder":
            _norm(out, "encoder.conv_norm_out", ch)
            _conv(out, "encoder.conv_out", 2 * z, ch, 3)
    for i, c in enumerate(boc[::-1]):
        for j in range(L + 1):
            _resnet(out, f"decoder.up_blocks.{i}.resnets.{j}", ch, c, 0)
            ch = c
        if i < len(boc) - 1:
            _conv(out, f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    _norm(out, "decoder.conv_norm_out", ch)
    _conv(out, "decoder.conv_out", cfg["out_channels"], ch, 3)
    _conv(out, "quant_conv", 2 * z, 2 * z, 1)
    _conv(out, "post_quant_conv", z, z, 1)
    return out


def numel(m):
    t = 0
    for _, s in m:
        n = 1
        for d in s:
            n *= d
        t += n
    return t


def write(path, m):
    with open(path, "w") as f:
        for k, s in sorted(m):
            f.write(f"{k} {'x'.join(str(d) for d in s)}\n")


def read(path):
    out = {}
    with open(path) as f:
        for line in f:
            k, s = line.split()
            out[k] = tuple(int(d) for d in s.split("x"))
    return out


if __name__ == "__main__":
    u, v = unet_manifest(), vae_manifest()
    assert numel(u) == 2_567_463_684, numel(u)
    assert numel(v) == 83_653_863, numel(v)
    assert len(set(k for k, _ in u)) == len(u) and len(set(k for k, _ in v)) == len(v)
    write(os.path.join(GOLDEN, "sdxl_unet_manifest.txt"), u)
    write(os.path.join(GOLDEN, "sdxl_vae_manifest.txt"), v)
    print(len(u), "UNet tensors,", len(v), "VAE tensors")
