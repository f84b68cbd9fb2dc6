"""CPU (no GPU): the C-ABI library loads and exports every symbol include/imh.h declares; the
host-side recording logic (op sequence, buffer lifetimes, FLOP accounting) of the fused UNet
forward; the lane-level emulator of the kernels' index math.  No compute call is made."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from imagharmony_amd import lib
    l = lib.load()
    hdr = open(os.path.join(ROOT, "include", "imh.h")).read()
    declared = set(re.findall(r"\b(imh_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    bound = {n for n, _, _ in lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for n in declared:
        assert hasattr(l, n)
    assert l.imh_abi_version() == lib.ABI_VERSION == int(re.search(r"#define IMH_ABI_VERSION (\d+)", hdr).group(1))


def test_ctypes_structs_match_header_field_order():
    from imagharmony_amd import lib
    hdr = open(os.path.join(ROOT, "include", "imh.h")).read()
    for cname, struct in (("imh_gemm_args", lib.GemmArgs), ("imh_attn_args", lib.AttnArgs),
                          ("imh_norm_args", lib.NormArgs), ("imh_ew_args", lib.EwArgs),
                          ("imh_small_attn_args", lib.SmallAttnArgs), ("imh_xattn_args", lib.XAttnArgs), ("imh_f32_args", lib.F32Args)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.replace("*", " ").split(",")
            names.append(parts[0].split()[-1])
            names.extend(p.strip() for p in parts[1:])
        assert names == [f[0] for f in struct._fields_], cname


def test_argument_errors_are_status_codes_not_crashes():
    from imagharmony_amd import lib
    l = lib.load()
    a = lib.GemmArgs()
    assert l.imh_gemm(a, None) == -1 and b"null" in l.imh_last_error()
    assert l.imh_plan_replay(None, None) == -1
    p = l.imh_plan_create()
    assert l.imh_plan_size(p) == 0
    a.X = a.W = a.Y = 64
    a.M, a.N, a.K = 8, 8, 64
    assert l.imh_plan_add(p, lib.OP_GEMM, lib.C.byref(a), 0, 7) == 0
    assert l.imh_plan_size(p) == 1 and l.imh_plan_get_tag(p, 0) == 7 and l.imh_plan_get_kind(p, 0) == lib.OP_GEMM
    assert l.imh_plan_add(p, 99, lib.C.byref(a), 0, 0) == -1
    l.imh_plan_destroy(p)
    bm, bn, sp = lib.C.c_int(), lib.C.c_int(), lib.C.c_int()
    assert l.imh_gemm_pick_config(2048, 1280, 11520, lib.C.byref(bm), lib.C.byref(bn), lib.C.byref(sp)) == 0
    assert bm.value in (64, 128) and bn.value in (64, 128) and sp.value >= 1
    assert l.imh_gemm_workspace_bytes(100, 200, 4) == 100 * 200 * 4 * 4


def test_product_path_has_no_cpu_fallback():
    from imagharmony_amd import lib
    from imagharmony_amd.ctx import Ctx
    with pytest.raises(lib.ImhError):
        Ctx("cpu", torch.bfloat16)
    src = ""
    for f in os.listdir(os.path.join(ROOT, "imagharmony_amd")):
        if f.endswith(".py"):
            src += open(os.path.join(ROOT, "imagharmony_amd", f)).read()
    assert "import oracle" not in src and "from oracle" not in src


def test_emulator_of_kernel_index_math():
    exe = os.path.join(ROOT, "tests", "emu", "emu_layout")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "emu", "emu_layout.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "FAIL" not in r.stdout


def test_unet_recording_dry_run_tiny():
    """records the whole forward on CPU tensors (never executed): op count, every pool buffer is
    released exactly once or still owned, processors install through the reference's dict protocol"""
    from imagharmony_amd.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
    from imagharmony_amd.ctx import Ctx
    from imagharmony_amd.unet import StepState, UNet2DConditionModel, UNetConfig
    cfg = UNetConfig(block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2),
                     attention_head_dim=(1, 2, 4), cross_attention_dim=256, addition_time_embed_dim=64,
                     projection_class_embeddings_input_dim=128 + 6 * 64, sample_size=32)
    u = UNet2DConditionModel(cfg).to(torch.bfloat16)
    procs = {}
    for name in u.attn_processors:                       # ip_adapter/ip_adapter.py:99-125
        if name.endswith("attn1.processor"):
            procs[name] = AttnProcessor2_0()
        else:
            hidden = 256 if name.startswith("mid_block") else (
                list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])] if name.startswith("up_blocks")
                else cfg.block_out_channels[int(name[len("down_blocks.")])])
            procs[name] = IPAttnProcessor2_0(hidden, 256, num_tokens=4,
                                            skip="down_blocks.2.attentions.1" not in name).to(torch.bfloat16)
    u.set_attn_processor(procs)
    assert len(u.attn_processors) == 2 * (2 + 4 + 2 + 6 + 3)
    ctx = Ctx("cpu", torch.bfloat16, record=True, dry=True)
    st = u.prepare_conditioning(ctx, torch.zeros(2, 81, 256), torch.zeros(2, 128), torch.zeros(2, 6))
    n_prep = ctx.lib.imh_plan_size(ctx.plan)
    st.t_value = torch.zeros(2)
    st.latents = torch.zeros(2, 4, 32, 32)
    out = u.emit_forward(ctx, st, 2, 32, 32, cfg_dup=False)
    assert out.shape == (2, 32 * 32, 4)
    n = ctx.lib.imh_plan_size(ctx.plan) - n_prep
    n_blocks = 2 + 4 + 2 + 6 + 3
    assert n > 12 * n_blocks
    kinds = [k for _, k, *_ in ctx.tags]
    from imagharmony_amd import lib as L
    assert kinds.count(L.OP_ATTN) == n_blocks                   # one self-attention launch per block ...
    assert kinds.count(L.OP_XATTN) == n_blocks                  # ... and one fused to_q + cross-attention launch
    assert kinds.count(L.OP_LAYERNORM) == 0                     # norm1/2/3 are folded into their consumer GEMMs
    with pytest.raises(Exception):
        ctx.run()


def test_unet_checkpoint_roundtrip_with_oracle_schema(tmp_path):
    """a diffusers-schema safetensors file written from the ORACLE UNet loads strictly into the HIP UNet holder
    (same key schema), and the IP-adapter state dict keeps the reference's '<idx>.to_k_ip.weight' keys"""
    from safetensors.torch import save_file
    from imagharmony_amd.ip_adapter import install_ip_processors
    from imagharmony_amd.unet import UNet2DConditionModel, UNetConfig
    from oracle.detfill import det_fill
    from oracle.sdxl_unet import UNet2DConditionModel as OracleUNet, tiny_config
    ocfg = tiny_config()
    ou = det_fill(OracleUNet(ocfg), 5)
    path = str(tmp_path / "unet.safetensors")
    save_file({k: v.contiguous() for k, v in ou.state_dict().items()}, path)
    cfg = UNetConfig(**{k: getattr(ocfg, k) for k in UNetConfig.__dataclass_fields__})
    hu = UNet2DConditionModel.from_safetensors(path, cfg)
    for k, v in ou.state_dict().items():
        assert torch.equal(hu.state_dict()[k], v), k
    procs = install_ip_processors(hu, num_tokens=4, dtype=torch.float32)
    ks = list(torch.nn.ModuleList(hu.attn_processors.values()).state_dict().keys())
    assert ks[0] == "1.to_k_ip.weight" and len(ks) == 2 * sum(k.endswith("attn2.processor") for k in procs)
    bad = {"conv_in.weight": torch.zeros(1)}
    save_file(bad, str(tmp_path / "bad.safetensors"))
    with pytest.raises(KeyError):
        UNet2DConditionModel.from_safetensors(str(tmp_path / "bad.safetensors"), cfg)


def test_header_enums_match_python_constants():
    """IMH_EW_* / IMH_GF_* / IMH_OP_* values in include/imh.h == the constants imagharmony_amd/lib.py hands to the ABI"""
    from imagharmony_amd import lib
    hdr = open(os.path.join(ROOT, "include", "imh.h")).read()
    vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(IMH_(?:EW|OP|GF)_[A-Z0-9_]+)\s*=\s*(\d+)", hdr)}
    vals.update({m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(IMH_(?:EW|OP|GF)_[A-Z0-9_]+)\s+(\d+)", hdr)})
    assert len([k for k in vals if k.startswith("IMH_EW_")]) == 12
    for k, v in vals.items():
        py = k[len("IMH_"):]
        if hasattr(lib, py):
            assert getattr(lib, py) == v, k
    for name in ("EW_TIMESTEP", "EW_CONV_IN", "EW_CFG_STEP", "EW_STEP_SET", "EW_CFG_RESCALE", "EW_SOFTMAX", "EW_ROW_STATS"):
        assert vals["IMH_" + name] == getattr(lib, name)


def test_vae_host_plumbing_matches_oracle_on_cpu(tmp_path):
    """the tile blends of the HIP VAE's tiled_decode are host-side torch: they must equal diffusers' in-place
    blend_v / blend_h (restated in oracle.vae), and the holder must take an oracle / diffusers state dict strictly"""
    from imagharmony_amd.vae import AutoencoderKL, VAEConfig, postprocess
    from oracle.detfill import det_fill, det_randn
    from oracle.vae import AutoencoderKL as OracleVAE
    from oracle.vae import postprocess as oracle_post
    from oracle.vae import tiny_vae_config
    a, b = det_randn((2, 3, 40, 24), 1), det_randn((2, 3, 40, 24), 2)
    for dim, fn in ((2, OracleVAE.blend_v), (3, OracleVAE.blend_h)):
        for extent in (8, 64):
            want = fn(a.clone(), b.clone(), extent)
            got = AutoencoderKL._blend(a.clone(), b.clone(), extent, dim)
            assert torch.allclose(got, want, atol=1e-6), (dim, extent)
    ocfg = tiny_vae_config()
    ov = det_fill(OracleVAE(ocfg), 3)
    hv = AutoencoderKL(VAEConfig(**{k: getattr(ocfg, k) for k in VAEConfig.__dataclass_fields__}))
    missing, unexpected = hv.load_state_dict(ov.state_dict(), strict=True)
    assert not missing and not unexpected
    assert torch.equal(hv.decoder.mid_block.attentions[0].to_q.weight, ov.decoder.mid_block.attentions[0].to_q.weight)
    assert hv.tile_latent_min_size == ov.tile_latent_min_size == 32
    img = det_randn((1, 3, 16, 16), 4) * 2
    assert np.array_equal(postprocess(img, "np"), oracle_post(img, "np"))
    assert np.array_equal(np.asarray(postprocess(img, "pil")[0]), np.asarray(oracle_post(img, "pil")[0]))
    with pytest.raises(ValueError):
        postprocess(img, "jpeg")


def test_pipeline_argument_checks_need_no_gpu():
    """eta != 0 is refused (the device-resident step is the deterministic update)"""
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    pipe = StableDiffusionXLCustomPipeline.__new__(StableDiffusionXLCustomPipeline)
    pipe.vae = pipe.vae_decode = None
    with pytest.raises(NotImplementedError, match="eta"):
        pipe(prompt_embeds=torch.zeros(1, 81, 8), eta=0.5, output_type="latent")
    # the reference's default output_type is "pil" (custom_pipelines.py:42): without a VAE that is refused up front
    with pytest.raises(NotImplementedError, match="needs a VAE"):
        pipe(prompt_embeds=torch.zeros(1, 81, 8))
    import inspect
    assert inspect.signature(StableDiffusionXLCustomPipeline.__call__).parameters["output_type"].default == "pil"


def test_no_packed_fp32_odd_register_selects_in_the_built_library():
    """the packed-fp32 hazard guard (csrc/imh_common.h IMH_KERNEL, tools/pk_fma_probe.py): no compiler-formed
    v_pk_{fma,mul,add}_f32 whose low lane reads the odd register of a pair may exist in any gfx950 code object"""
    import shutil
    from imagharmony_amd import lib
    from tools.check_packed_selects import LLVM, offenders
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    bad, n_objs, n_pk = offenders(lib.LIB_PATH)
    assert n_objs >= 6 and n_pk > 1000, (n_objs, n_pk)        # the disassembly really saw the kernels
    assert not bad, bad[:5]


def test_integration_md_gemm_stub_matches_lib_and_header():
    """the ctypes stub INTEGRATION.md shows integrators is the struct the library really takes (field names and order
    equal imagharmony_amd.lib.GemmArgs, which test_ctypes_structs_match_header_field_order ties to include/imh.h)"""
    import re
    from imagharmony_amd import lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md[md.index("class GemmArgs(C.Structure):"):md.index("lib.imh_gemm.argtypes")]
    fields = re.findall(r'\("(\w+)",\s*C\.(\w+)\)', block)
    import ctypes as C
    want = list(lib.GemmArgs._fields_)
    assert [f[0] for f in fields] == [w[0] for w in want]
    assert [getattr(C, f[1]) for f in fields] == [w[1] for w in want]       # c_int32 is c_int on this ABI: compare the types


def test_every_tuning_entry_names_a_variant_the_dispatcher_knows():
    """tuning.json is written by a sweep on the GPU box; a (bm, bn) pair the C++ dispatch does not list would only fail
    at launch time there.  The dispatch tables are plain `bm == A && bn == B` chains: read them from the sources."""
    import json
    import re
    csrc = os.path.join(ROOT, "imagharmony_amd", "csrc")
    known = set()
    for f in ("gemm.hip", "gemm_ring.hip"):
        for a, b in re.findall(r"bm == (\d+) && bn == (\d+)", open(os.path.join(csrc, f)).read()):
            known.add((int(a), int(b)))
    pp = {8256: 256, 9128: 320, 9256: 320}                         # gemm_pp.hip: one tile shape per code
    halo = {(7128, 320), (7128, 160), (7128, 80), (7564, 320), (7564, 160), (7328, 160), (7428, 160), (7256, 160), (7356, 160)}    # conv_halo.hip: patch x couts, stride-1 conv only
    table = json.load(open(os.path.join(ROOT, "imagharmony_amd", "tuning.json")))
    assert len(table) >= 30
    for key, (bm, bn, splits) in table.items():
        M, N, K, conv = (int(v) for v in key.split(",")[:4])      # (an optional fifth field 1: the entry for precomputed LN statistics)
        assert splits >= 1 and K % 64 == 0, key
        if (bm, bn) in halo:
            assert conv == 1, key
        elif bm in pp:
            assert pp[bm] == bn and conv == 0 and splits == 1, key
        else:
            assert (bm, bn) in known, f"{key}: variant {bm} x {bn} is not in the dispatch tables"
        # ... and it must be one the DEFAULT library compiles: the measured-but-not-selected variants exist only with -DIMH_EXPERIMENTAL
        from imagharmony_amd import lib as L
        assert (bm, bn) not in L.EXP_VARIANTS, f"{key}: variant {bm} x {bn} is an experimental-only variant"


def test_folded_layernorm_launches_only_get_variants_that_implement_it():
    """host logic of Ctx.gemm: a tuning-table variant without the folded LayerNorm (rings, KG2, ping-pong) is replaced by a plain
    tile when the launch carries ln=..., the variants that have it (plain tiles, every wave-specialised one) are kept, an
    explicit cfg is passed through; a launch without statistics from its caller gets them from a row-statistics op emitted in
    front of it (the kernels have no in-loop form), and the C ABI itself refuses a folded launch without ln_stats"""
    from imagharmony_amd import lib as L
    from imagharmony_amd.ctx import Ctx
    ctx = Ctx("cpu", torch.bfloat16, record=True, dry=True)
    M, N, K = 2048, 1280, 2560
    x, w = torch.zeros(M, K, dtype=torch.bfloat16), torch.zeros(N, K, dtype=torch.bfloat16)
    s, c = torch.zeros(N), torch.zeros(N)
    st = (torch.zeros(M, K // 80, 2), K // 80)
    for table, want in (((1464, 160, 1), (1464, 160)), ((256, 256, 1), (64, 64)), ((3128, 128, 1), (64, 64)), ((64, 128, 1), (64, 128)),
                        ((2464, 160, 1), (2464, 160)), ((23256, 160, 1), (23256, 160)), ((9128, 320, 1), (64, 64)), ((8256, 256, 1), (64, 64)),
                        ((64, 64, 1), (64, 64))):
        ctx.tuning[(M, N, K, 0)] = table
        a = ctx.gemm(x, w, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), _args_only=True)[0]
        assert (a.bm, a.bn, a.splits) == (*want, 1), (table, a.bm, a.bn, a.splits)
        assert a.ln_stats == st[0].data_ptr() and a.ln_slots == st[1]
        a = ctx.gemm(x, w, _args_only=True)[0]                      # without LN the table entry is used as is
        assert (a.bm, a.bn) == table[:2]
    a = ctx.gemm(x, w, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), cfg=(9128, 320, 1), _args_only=True)[0]
    assert (a.bm, a.bn) == (9128, 320)                              # explicit: the C side is the one to refuse it
    # no statistics from the caller: a row-statistics op over the token rows goes in front, its output feeds the launch
    n0 = len(ctx.tags)
    ctx.gemm(x, w, flags=L.GF_LN_ROW, ln=(s, c, 1e-5))
    kinds = [(t[1], t[2]) for t in ctx.tags[n0:]]
    assert kinds == [(L.OP_EW, "gemm.ln_row_stats"), (L.OP_GEMM, "gemm")], kinds
    assert ctx._ops[-1][1].ln_stats == ctx._ops[-2][1].y and ctx._ops[-1][1].ln_slots == 1
    n0 = len(ctx.tags)
    ctx.gemm(w, x, flags=L.GF_LN_COL, ln=(torch.zeros(N), torch.zeros(N), 1e-5))    # column form: the tokens are the W operand's rows
    assert ctx._ops[-2][1].a == x.data_ptr() and ctx._ops[-1][1].ln_stats == ctx._ops[-2][1].y
    with pytest.raises(L.ImhError, match="stats_out"):
        ctx.gemm(x, w, flags=L.GF_OUT_F32, stats_out=True)
    with pytest.raises(L.ImhError, match="mutually exclusive"):
        ctx.gemm(x, w, stats_out=True, gn_out=(32, 1024))
    # the C ABI without statistics
    a = ctx.gemm(x, w, flags=L.GF_LN_ROW, ln=(s, c, 1e-5, st), cfg=(128, 128, 1), _args_only=True)[0]
    a.ln_stats = None
    assert ctx.lib.imh_gemm(a, None) == -1 and b"ln_stats" in ctx.lib.imh_last_error()


def test_groupnorm_statistics_handover_host_logic():
    """host logic of the GroupNorm hand-over (Ctx._gn_epilogue / gn_table / conv3x3(gn=..., x2=...)): a producing launch gets gn_out
    only when its tile variant has the epilogue AND the shape fits it (N % 10 == 0, whole pixel blocks, no split-K); the table op
    carries one or two producers' buffers; the fused front end and the two-source input are only handed to LDS-halo launches"""
    from imagharmony_amd import lib as L
    from imagharmony_amd.ctx import Ctx, GnStats
    ctx = Ctx("cpu", torch.bfloat16, record=True, dry=True)
    bf = torch.bfloat16
    x, w = torch.zeros(2, 32, 32, 64, dtype=bf), torch.zeros(320, 9 * 64, dtype=bf)
    for cfg, rows in (((7128, 320, 1), 32), ((7564, 160, 1), 16), ((7256, 160, 1), 64), ((2464, 160, 1), 32), ((23256, 160, 1), 64),
                      ((128, 128, 1), 0), ((2464, 160, 2), 0), ((24128, 128, 1), 0)):
        assert ctx.lib.imh_gemm_gn_block_rows(cfg[0], cfg[1]) == (rows if cfg != (2464, 160, 2) else 32)
        y, gs = ctx.conv3x3(x, w, cfg=cfg, gn_groups=32)
        epi = ctx.tags[-1][6]
        if rows:
            assert isinstance(gs, GnStats) and gs.nblk == 1024 // rows and tuple(gs.t.shape) == (2, 1024 // rows, 32, 2) and gs.t.dtype == torch.float32
            assert (gs.sub, gs.npart, gs.C) == (10, 10 * rows, 320) and epi["gn_out"] == (1024 // rows, 1024)
            n0 = len(ctx.tags)
            ctx.groupnorm(y.view(2, 1024, 320), None, None, 32, 1e-5, True, stats=gs)
            a_op = ctx._ops[-1][1]                                      # ONE launch: the apply pass builds its sample's table itself
            assert [t[2] for t in ctx.tags[n0:]] == ["groupnorm"]       # (no statistics pass, no table launch)
            assert (a_op.mode, a_op.partial, a_op.nblk, a_op.sub, a_op.npart) == (L.GN_TABLE_APPLY, gs.t.data_ptr(), gs.nblk, 10, 10 * rows)
            assert a_op.silu == 1 and a_op.groups == 32 and not a_op.table
            tab_ = ctx.gn_table(gs, None, None, 32, 1e-5, 1024)         # the stand-alone steps are still there (consumers that share a table)
            ctx.gn_apply(y.view(2, 1024, 320), tab_, True)
            t_op, p_op = ctx._ops[-2][1], ctx._ops[-1][1]
            assert (t_op.mode, t_op.partial, t_op.nblk, t_op.sub, t_op.npart) == (L.GN_TABLE, gs.t.data_ptr(), gs.nblk, 10, 10 * rows)
            assert p_op.mode == L.GN_APPLY and p_op.table == t_op.table and p_op.silu == 1
        else:
            assert gs is None and epi["gn_out"] is None
            ctx.groupnorm(y.view(2, 1024, 320), None, None, 32, 1e-5, True)
            assert ctx._ops[-1][1].mode == L.GN_ALL
    # shapes off the grid: N not a multiple of 10, a patch grid that does not tile the image, rows per sample not a block multiple
    assert ctx.conv3x3(x, torch.zeros(384, 9 * 64, dtype=bf), cfg=(7128, 160, 1), gn_groups=32)[1] is None
    assert ctx.conv3x3(torch.zeros(1, 12, 20, 64, dtype=bf), w, cfg=(7128, 320, 1), gn_groups=32)[1] is None
    xg, wg = torch.zeros(2 * 48, 64, dtype=bf), torch.zeros(320, 64, dtype=bf)
    assert ctx.gemm(xg, wg, cfg=(23256, 160, 1), gn_out=(32, 48))[1] is None
    assert ctx.gemm(torch.zeros(512, 64, dtype=bf), wg, cfg=(23256, 160, 1), gn_out=256)[1].nblk == 4
    with pytest.raises(L.ImhError, match="statistics"):
        ctx.groupnorm(torch.zeros(2, 1024, 320, dtype=bf), None, None, 32, 1e-5, True, stats=GnStats(torch.zeros(2, 8, 16, 2), 8, 10, 320, 160))
    # two producers (channel concat 640 + 320 -> 30 channels per group) -> one table; the fused conv reads both sources
    a_, b_ = torch.zeros(2, 32, 32, 640, dtype=bf), torch.zeros(2, 32, 32, 320, dtype=bf)
    ga, gb = ctx.gn_stats(a_.view(2, 1024, 640)), ctx.gn_stats(b_.view(2, 1024, 320))
    tab = ctx.gn_table([ga, gb], None, None, 32, 1e-5, 1024)
    t_op = ctx._ops[-1][1]
    assert tuple(tab.shape) == (2, 960, 2) and (t_op.C, t_op.C1, t_op.partial2, t_op.sub2, t_op.npart, t_op.npart2) == (960, 640, gb.t.data_ptr(), 10, 0, 0)
    w9 = torch.zeros(640, 9 * 960, dtype=bf)
    assert ctx.conv_fuses_gn(2048, 640, 9 * 960, cfg=(7128, 160, 1)) and not ctx.conv_fuses_gn(2048, 640, 9 * 960, cfg=(2464, 160, 1))
    assert not ctx.conv_fuses_gn(2048, 640, 9 * 960, stride=2, cfg=(7128, 160, 1))
    ctx.conv3x3(a_, w9, cfg=(7128, 160, 1), gn=(tab, True), x2=b_)
    c_op = ctx._ops[-1][1]
    assert (c_op.X, c_op.X2, c_op.Cin, c_op.Cin1, c_op.gn_tab, c_op.gn_silu) == (a_.data_ptr(), b_.data_ptr(), 960, 640, tab.data_ptr(), 1)
    # ... or the partials themselves (round 5: the conv builds its sample's table in its prologue, no table launch)
    from imagharmony_amd.ctx import GnSpec
    ctx.conv3x3(a_, w9, cfg=(7128, 160, 1), gn=(GnSpec([ga, gb], None, None, 32, 1e-5), True), x2=b_)
    s_op = ctx._ops[-1][1]
    assert not s_op.gn_tab and (s_op.gn_part, s_op.gn_part2, s_op.gn_pC1, s_op.gn_groups, s_op.gn_silu) == (ga.t.data_ptr(), gb.t.data_ptr(), 640, 32, 1)
    assert (s_op.gn_pnblk, s_op.gn_psub, s_op.gn_pnpart, s_op.gn_pnblk2, s_op.gn_psub2) == (ga.nblk, 10, 0, gb.nblk, 10) and abs(s_op.gn_eps - 1e-5) < 1e-12
    with pytest.raises(L.ImhError, match="statistics"):
        ctx.conv3x3(a_, w9, cfg=(7128, 160, 1), gn=(GnSpec([ga], None, None, 32, 1e-5), True), x2=b_)
    with pytest.raises(L.ImhError, match="LDS-halo"):
        ctx.conv3x3(a_, w9, cfg=(2464, 160, 1), gn=(tab, True), x2=b_)
    with pytest.raises(L.ImhError, match="table"):
        ctx.conv3x3(a_, w9, cfg=(7128, 160, 1), gn=(torch.zeros(2, 640, 2), True), x2=b_)
    # the shortcut GEMM over the same two sources; a variant that cannot read them is replaced (or refused when explicit)
    wsc = torch.zeros(640, 960, dtype=bf)
    ctx.tuning[(2048, 640, 960, 0)] = (9128, 320, 1)
    g_op = ctx.gemm(a_.view(2048, 640), wsc, x2=b_.view(2048, 320), _args_only=True)[0]
    assert (g_op.K, g_op.Cin1, g_op.X2) == (960, 640, b_.data_ptr()) and g_op.bm <= 128
    ctx.tuning[(2048, 640, 960, 0)] = (24128, 160, 1)
    assert ctx.gemm(a_.view(2048, 640), wsc, x2=b_.view(2048, 320), _args_only=True)[0].bm == 24128


def test_derived_weight_caches_follow_in_place_updates():
    """packed / LayerNorm-folded weight copies are cached on the modules; their keys carry the in-place version counter of
    every source tensor (weight, bias, norm.weight, norm.bias), so load_state_dict / weight.copy_ after a first forward
    rebuild them -- the storage address alone does not change on an in-place update"""
    from imagharmony_amd.attention_processor import _packed_qk, _cached, _vkey, fold_ln
    from imagharmony_amd.ctx import Ctx
    from imagharmony_amd.unet import Attention, Conv2d, GEGLU, Norm
    ctx = Ctx("cpu", torch.bfloat16, record=True, dry=True)
    g = torch.Generator().manual_seed(0)
    attn = Attention(64, 1)
    norm, ge, conv = Norm(64, 1e-5), GEGLU(64, 128), Conv2d(64, 64, 3)
    with torch.no_grad():
        for p in list(attn.parameters()) + list(norm.parameters()) + list(ge.parameters()) + list(conv.parameters()):
            p.copy_(torch.randn(p.shape, generator=g))
    qk0 = _packed_qk(attn, ctx).clone()
    assert _packed_qk(attn, ctx).data_ptr() == _packed_qk(attn, ctx).data_ptr()          # cached while nothing changes
    w0, b0, s0, c0 = [t.clone() if t is not None else None for t in ge.packed_ln(ctx, norm)]
    assert b0 is None                    # the Linear's bias rides in the fold's constant term c = W beta + b
    cw0 = conv.packed(ctx).clone()
    key = lambda: (_vkey(attn.to_q.weight, norm.weight, norm.bias), ctx.dtype, str(ctx.device))
    f0 = [t.clone() for t in _cached(attn, "_imh_ln_q", key(), lambda: fold_ln(attn.to_q.weight, norm, ctx))]
    with torch.no_grad():                 # in-place updates: same data_ptr, new version
        attn.to_q.weight.mul_(2.0)
        norm.bias.add_(1.0)
        conv.weight.mul_(0.5)
    assert not torch.equal(_packed_qk(attn, ctx), qk0)
    w1, b1, s1, c1 = ge.packed_ln(ctx, norm)
    assert torch.equal(w1, w0) and not torch.equal(c1, c0)                               # beta moved: c = W beta follows
    assert torch.equal(conv.packed(ctx), (cw0.float() * 0.5).to(torch.bfloat16))
    f1 = _cached(attn, "_imh_ln_q", key(), lambda: fold_ln(attn.to_q.weight, norm, ctx))
    assert not torch.equal(f1[0], f0[0]) and not torch.equal(f1[2], f0[2])
    sd = {k: v.clone() for k, v in attn.state_dict().items()}
    qk1 = _packed_qk(attn, ctx).clone()
    sd["to_k.weight"] = sd["to_k.weight"] + 1.0
    attn.load_state_dict(sd)              # load_state_dict copies in place
    assert not torch.equal(_packed_qk(attn, ctx), qk1)

def test_inline_asm_vmem_stores_carry_their_hazard_nop():
    """hipcc's hazard recognizer does not look inside inline asm: a global_store of more than 64 bits issued from asm still reads its data
    registers when the next instruction overwrites them (round 5: an LDS-staged epilogue built on such a store returned corrupted lines
    and looked 1.5 ms per forward faster).  Every inline-asm VMEM store in the kernel sources must be followed by an s_nop in the same
    asm statement."""
    import re
    csrc = os.path.join(ROOT, "imagharmony_amd", "csrc")
    found = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        src = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r'asm volatile\("(global_store_dwordx[24][^"]*)"', src):
            found += 1
            assert "s_nop" in m.group(1), f"{f}: inline-asm store without its hazard nop: {m.group(1)[:80]}"
    assert found >= 1
