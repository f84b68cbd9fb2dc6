"""GPU tool: A/B of whole-forward configurations in ONE process on one box (guide rule 24: perf deltas come from interleaved
rounds inside one process).  Records the 1024^2 CFG-2 SDXL forward once per configuration, then times the recorded plans
in interleaved rounds: per-op HIP-event times (min over rounds, summed by op family) and the back-to-back wall time of the
plan (median over rounds).  Knobs per configuration:
    ln_stats   unet.LN_STATS_HANDOVER (LayerNorm statistics handed over from the producing GEMM's epilogue)
    gn_stats   unet.GN_STATS_HANDOVER (GroupNorm statistics likewise)
    gn_fuse    unet.GN_FUSE (GroupNorm + SiLU + channel concat inside the LDS-halo convs)
    attn       imh_debug_set(4, mode): self-attention key loop
    xattn      imh_debug_set(3, mode): 1 one head per workgroup, 2 / 3 / 4 two heads with 0 / 2 / 4 producer waves
    tuning     {"M,N,K,conv[,1]": [bm, bn, splits]} overrides on top of tuning.json
Usage: python tools/forward_ab.py [--rounds 5] [--configs name1,name2,...] [--stacked S] > gpurun_out/forward_ab.json"""
import argparse
import collections
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagharmony_amd import lib as L                                   # noqa: E402
from imagharmony_amd import unet as U                                  # noqa: E402
from imagharmony_amd import attention_processor as AP                 # noqa: E402
from imagharmony_amd.ctx import _load_tuning                           # noqa: E402
from tools.sweep import DEV, build_unet, record                        # noqa: E402

from imagharmony_amd.ctx import Ctx as _CtxD                          # noqa: E402
_PF_DEFAULT = (_CtxD.PF_BIG, _CtxD.PF_BIG_CAP)
CONFIGS = collections.OrderedDict([
    # round 4 (the round-2 / round-3 sessions' configurations are in git history; their results in profiles/r03_forward_ab_*.json)
    ("base", dict()),
    # round 6: where does the tail prefetch of the 26-MB ff.net.0 weight go?  cap = only the first N MB prefetched (by the launch before); chunk = N-MB pieces
    # handed to the launches before, nearest first
    ("pf_cap8", dict(pf_chunk=8 << 20, pf_cap=True)), ("pf_cap4", dict(pf_chunk=4 << 20, pf_cap=True)), ("pf_off", dict(pf_chunk=4096, pf_cap=True)),
    ("w16_pf_tail", dict(w16_pf=0)),        # round 6: ff.net.0 prefetching ff.net.2 behind its epilogue (the in-loop form is the default)
    ("pf_everything", dict(pf_big=1 << 30, pf_big_cap=0)),        # round 5's rule: every weight rides the previous launch's tail
    ("pf_big0", dict(pf_big=14 << 20, pf_big_cap=0)), ("pf_big2", dict(pf_big=14 << 20, pf_big_cap=2 << 20)), ("pf_big4", dict(pf_big=14 << 20, pf_big_cap=4 << 20)),
    ("pf_big8", dict(pf_big=14 << 20, pf_big_cap=8 << 20)), ("pf_big13", dict(pf_big=14 << 20, pf_big_cap=13 << 20)),
    ("pf_chunk8", dict(pf_chunk=8 << 20, pf_back=4)), ("pf_chunk13", dict(pf_chunk=13 << 20, pf_back=3)), ("pf_chunk4", dict(pf_chunk=4 << 20, pf_back=7)),
    # round 6: ff.net.0 on the sixteen-wave 256 x 320 tile (gemm_w16.hip) -- one round of 256 workgroups at UNet batch 2
    ("geglu_w16", dict(tuning={"2048,10240,1280,0,1": [26256, 320, 1], "8192,5120,640,0,1": [26256, 320, 1],
                               "8192,10240,1280,0,1": [26256, 320, 1], "32768,5120,640,0,1": [26256, 320, 1]})),
    ("geglu_w16_1280", dict(tuning={"2048,10240,1280,0,1": [26256, 320, 1], "8192,10240,1280,0,1": [26256, 320, 1]})),
    # self-attention key loop (imh_debug_set key 4): 1 in-order, 3 software-pipelined + deferred maximum (the round-3 default),
    # 5 key-split workgroups (round 4)
    ("attn1", dict(attn=1)),
    ("attn3", dict(attn=3)),
    ("attn5", dict(attn=5)),
    ("attn_whole_items", dict(attn=7)),
    ("qkv_256x128", dict(tuning={"2048,3840,1280,0,1": [23256, 128, 1]})),      # the one-launch [Q|K|V] of the L = 1024 layers on 240 tiles
    ("qkv_256x128_all", dict(tuning={"2048,3840,1280,0,1": [23256, 128, 1], "8192,1920,640,0,1": [23256, 128, 1]})),      # the pipelined kernel without the key-quarter workgroups
    # LayerNorm statistics from stand-alone row-statistics launches instead of the producers' epilogues
    ("ln_rowstats", dict(ln_stats=False)),
    ("gn_off", dict(gn_stats=False)),
    # GroupNorm + SiLU + concat inside the LDS-halo convs (round 4) vs table + apply passes + materialised concats
    ("gn_unfused", dict(gn_fuse=False)),
    ("gn_fused", dict(gn_fuse=True)),
    ("gn_fused_w8", dict(gn_fuse=True, halo=1)),      # imh_debug_set key 5: 1 = the eight-wave conv form for the fused launches too
    ("halo_w12", dict(gn_fuse=True, halo=2)),         # 2 = four halo waves for every LDS-halo conv
    # self-attention projections: the two-problem launch of round 3 vs ONE wave-specialised [Q|K|V] launch (at both widths / at C = 1280 only)
    # round-4 tile checks (in situ): [Q|K|V] at C = 640 on 128-row tiles; conv_shortcut over the 960-channel concat on the
    # wave-specialised 128 x 160; the fused 64 x 64 convolutions on 320-cout tiles
    ("qkv640_24128", dict(tuning={"8192,1920,640,0,1": [24128, 160, 1]})),
    ("sc960_24128", dict(tuning={"32768,320,960,0": [24128, 160, 1], "32768,320,640,0": [24128, 160, 1]})),
    ("conv64_320", dict(tuning={f"8192,640,{k},1": [7128, 320, 1] for k in (2880, 5760, 8640, 11520, 17280)})),
    ("conv128_7128", dict(tuning={f"32768,320,{k},1": [7128, 320, 1] for k in (2880, 5760, 8640)})),
    # the 32 x 32 ResBlock convolutions on the 4 x 16-patch LDS-halo form (GroupNorm + SiLU + concat fused) instead of the
    # wave-specialised 64 x 160 implicit GEMM behind a table + apply pass
    ("conv32_7564", dict(tuning={f"2048,1280,{k},1": [7564, 160, 1] for k in (5760, 11520, 17280, 23040)})),
    # round 5: the same convolutions on the 8 x 16-patch x 80-cout form whose wave pairs split the K range (conv_halo.hip KS)
    ("conv32_ks80", dict(tuning={f"2048,1280,{k},1": [7128, 80, 1] for k in (5760, 11520, 17280, 23040)})),
    ("conv32_ws", dict(tuning={f"2048,1280,{k},1": [2464, 160, 1] for k in (5760, 11520, 17280, 23040)})),
    ("conv64_ks80", dict(tuning={f"8192,640,{k},1": [7128, 80, 1] for k in (2880, 5760, 8640, 11520, 17280)})),
    ("conv128_ks80", dict(tuning={f"32768,320,{k},1": [7128, 80, 1] for k in (2880, 5760, 8640)})),
    ("gn_table_launches", dict(gn_fold=False)),      # round 5: one gn_table launch per GroupNorm instead of the consumers building their sample's table
    ("conv64_p16ks80", dict(tuning={f"8192,640,{k},1": [7256, 80, 1] for k in (2880, 5760, 8640, 11520, 17280)})),      # 16 x 16 patch x 80 couts, K split
    ("conv64_p16ks80_w4", dict(halo=5, tuning={f"8192,640,{k},1": [7256, 80, 1] for k in (2880, 5760, 8640, 11520, 17280)})),
    ("conv128_p16ks80", dict(tuning={f"32768,320,{k},1": [7256, 80, 1] for k in (2880, 5760, 8640)})),
    ("res_late", dict(ws_early=0)),                   # imh_debug_set key 6 = 0: residual rows fetched after the K loop (rounds 2-4)
    ("ks_service_transform", dict(halo=8)),           # round 6: the 8 x 16 x 80 K-split form with the eight service waves transforming the whole halo (default: all sixteen waves share it)
    ("geglu_sixteen_waves", dict(w16_form=0)),        # round 6: the 256 x 320 ff.net.0 tile on sixteen 64 x 80 waves (default: eight fat waves of 128 x 80)
    ("b8_conv64_hws", dict(tuning={"32768,640,5760,1": [7356, 160, 1]})),      # round 6, UNet batch 8: the 64^2 K = 5760 convs on conv_hws (fused norm) instead of the 256 x 320 implicit GEMM
    ("b8_conv64_hws_7128", dict(tuning={"32768,640,5760,1": [7128, 160, 1]})),
    ("halo_lockstep", dict(halo=6)),                  # round 6: conv_halo.hip's kernels for the 16 x 16 / 8 x 16 patch x 160 forms (default: conv_hws.hip, wave-specialised)
    ("halo_svc8", dict(halo=4)),                      # eight service waves on the 8 x 16 x 160 forms (experimental build)
    ("halo_svc8_ring3", dict(halo=4, tuning={f"8192,640,{k},1": [7328, 160, 1] for k in (2880, 5760, 8640, 11520, 17280)})),
    ("halo_svc", dict(halo=3)),                       # imh_debug_set key 5 = 3: four halo waves for every LDS-halo conv, as SERVICE waves (they also run the weight ring)
    ("qkv_dual", dict(qkv_one=False)),
    ("qkv_one", dict(qkv_one=True)),
    ("qkv_one_1280", dict(qkv_one=True, qkv_widths=(1280,))),
    # the round-3 launch structure as far as it can still be selected: [Q|K] + V^T two-problem launch, GroupNorm as table + apply
    # passes in front of every conv (the erf-GELU polynomial and the statistics-only LayerNorm hand-over cannot be switched back)
    ("round3_like", dict(qkv_one=False, gn_fuse=False)),
    # (measured with this tool and removed from the tree again, results kept: the attention / fused cross-attention work items dealt to
    # the XCDs by query rows instead of by head, alone and with 8 x 1 cells on the Linears ("token rows stay on an XCD") --
    # profiles/r04_forward_ab_attention_items_by_rows_box*.json: the fused cross-attention 1-1.4 us faster per launch, the forward the
    # same to 0.01 ms; ff.net.0 as 256 PERSISTENT workgroups walking two tiles
    # each (the producers stream the next tile while the consumers are in the epilogue) -- profiles/r04_forward_ab_persistent_geglu.json:
    # bit-identical, 57.3 vs 57.6 us warm, 75.9 vs 64.6 us in the forward; ff.net.0 / [Q|K|V] on FOUR consumer waves of 128 x 80 with
    # streamed token fragments (29 % fewer LDS fragment bytes per K tile) -- profiles/r04_forward_ab_wave_tile_128x80.json: bit-identical,
    # [Q|K|V] the same, ff.net.0 68 -> 93 us; the halo waves of the LDS-halo conv as full producer
    # waves that also run the weight ring, and the MFMA waves' (chunk, tap) loop software-pipelined like the wave-specialised GEMM's
    # consumers -- profiles/r04_forward_ab_halo_producer_waves.json (8-11 % slower), r04_forward_ab_halo_pipelined_loop*.json (flat); the [Q|K|V] column tiles dealt to the XCDs by head
    # group so that Q, K, V of a head are written on the XCD that reads them -- profiles/r04_forward_ab_qkv_xcd_affinity.json, no
    # change; an L2 warm-up touch of the head's K / V^T lines at entry of the attention kernels --
    # profiles/r04_forward_ab_attn_kv_warmup.json, +1.8 / +0.8 us per launch)
    # XCD cell shape forced for EVERY GEMM / conv launch (imh_debug_set key 2): (8,1) (4,2) (2,4) (1,8) = M x N cells; read per op
    ("cells13", dict(cells=13)), ("cells3", dict(cells=3)), ("cells2", dict(cells=2)),      # Ctx.xcd_cells (the per-launch request, as DenoiseEngine picks it)
    ("base_again", dict()),                            # position control: the same configuration twice in one interleaved round
    ("xcd81", dict(xcd=2)), ("xcd42", dict(xcd=3)), ("xcd24", dict(xcd=4)), ("xcd18", dict(xcd=5)),
    ("xcd_m2", dict(xcd=6)), ("xcd_m3", dict(xcd=7)),   # the cost model restricted to (8,1) (4,2) / to (8,1) (4,2) (2,4)
    ("x1", dict(xattn=1)), ("x3", dict(xattn=3)), ("x4", dict(xattn=4)), ("x_whole_items", dict(xattn=9)),
])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--configs", default="base")
    ap.add_argument("--stacked", type=int, default=1)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    lib = L.load()
    u = build_unet(dtype)
    names = [n for n in a.configs.split(",") if n]
    plans = {}
    outs = {}
    for n in names:
        c = CONFIGS[n]
        U.LN_STATS_HANDOVER = bool(c.get("ln_stats", True))
        U.GN_STATS_HANDOVER = bool(c.get("gn_stats", True))
        U.GN_FUSE = bool(c.get("gn_fuse", True))
        U.GN_TABLE_FOLD = bool(c.get("gn_fold", True))
        AP.DUAL_WS = bool(c.get("dual_ws", False))
        AP.QKV_ONE = bool(c.get("qkv_one", True))
        AP.QKV_ONE_WIDTHS = tuple(c.get("qkv_widths", (640, 1280)))
        lib.imh_debug_set(3, int(c.get("xattn", 0)))
        lib.imh_debug_set(4, int(c.get("attn", 0)))
        lib.imh_debug_set(2, int(c.get("xcd", 0)))
        lib.imh_debug_set(5, int(c.get("halo", 0)))
        lib.imh_debug_set(6, int(c.get("ws_early", 1)))
        lib.imh_debug_set(7, int(c.get("w16_pf", 1)))
        lib.imh_debug_set(9, int(c.get("w16_form", 1)))
        from imagharmony_amd.ctx import Ctx as _Ctx
        _Ctx.PF_CHUNK, _Ctx.PF_CAP_ONLY, _Ctx.PF_BACK = int(c.get("pf_chunk", 0)), bool(c.get("pf_cap", False)), int(c.get("pf_back", 3))
        _Ctx.PF_BIG, _Ctx.PF_BIG_CAP = int(c.get("pf_big", _PF_DEFAULT[0])), int(c.get("pf_big_cap", _PF_DEFAULT[1]))
        tun = dict(_load_tuning())
        for k, v in (c.get("tuning") or {}).items():
            tun[tuple(int(x) for x in k.split(","))] = tuple(v)
        rec, out, st = record(u, dtype, 128, S=a.stacked, tuning=tun, cells=int(c.get("cells", 0)))
        rec.run()
        torch.cuda.synchronize()
        _Ctx.PF_CHUNK, _Ctx.PF_CAP_ONLY, _Ctx.PF_BACK = 0, False, 3
        _Ctx.PF_BIG, _Ctx.PF_BIG_CAP = _PF_DEFAULT
        plans[n] = (rec, c)
        outs[n] = out.float().clone()
    ref = outs[names[0]]
    res = {n: dict(cfg=CONFIGS[n], wall_ms=[], per_op=None,
                   rel_rms_vs_first=float(((outs[n] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())) for n in names}
    for r in range(a.rounds):
        for n in names:
            rec, c = plans[n]
            lib.imh_debug_set(3, int(c.get("xattn", 0)))
            lib.imh_debug_set(4, int(c.get("attn", 0)))
            lib.imh_debug_set(2, int(c.get("xcd", 0)))
            lib.imh_debug_set(5, int(c.get("halo", 0)))
            lib.imh_debug_set(6, int(c.get("ws_early", 1)))
            lib.imh_debug_set(7, int(c.get("w16_pf", 1)))
            lib.imh_debug_set(9, int(c.get("w16_form", 1)))
            ms = rec.time_ops()
            res[n]["per_op"] = ms if res[n]["per_op"] is None else [min(x, y) for x, y in zip(res[n]["per_op"], ms)]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rec.run()
            e1.record()
            torch.cuda.synchronize()
            res[n]["wall_ms"].append(e0.elapsed_time(e1) / 3)
    lib.imh_debug_set(3, 0)
    lib.imh_debug_set(4, 0)
    lib.imh_debug_set(2, 0)
    lib.imh_debug_set(5, 0)
    lib.imh_debug_set(6, 1)
    lib.imh_debug_set(7, 1)
    lib.imh_debug_set(9, 1)
    out = {}
    for n in names:
        rec, c = plans[n]
        by = collections.OrderedDict()
        for (tag, kind, descr, fl, by_, *rest), t in zip(rec.tags, res[n]["per_op"]):
            d = by.setdefault(descr, dict(n=0, ms=0.0, gflop=0.0))
            d["n"] += 1; d["ms"] += t; d["gflop"] += fl / 1e9
        shp = collections.OrderedDict()          # GEMM / conv launches by (op, shape, variant): the floor model's input (tools/floor_model.py)
        for (tag, kind, descr, fl, by_, *rest), t in zip(rec.tags, res[n]["per_op"]):
            shape, epi = (rest + [None, None])[:2]
            if kind != L.OP_GEMM or shape is None or not epi or "cfg" not in epi:
                continue
            k2 = (descr, tuple(shape[:4]), tuple(epi["cfg"]))
            d2 = shp.setdefault(k2, dict(n=0, ms=0.0))
            d2["n"] += 1; d2["ms"] += t
        by_shape = [dict(op=k2[0], M=k2[1][0], N=k2[1][1], K=k2[1][2], conv=k2[1][3], cfg=list(k2[2]), n=d2["n"], us_each=1e3 * d2["ms"] / d2["n"])
                    for k2, d2 in shp.items()]
        out[n] = dict(cfg=c, rel_rms_vs_first=res[n]["rel_rms_vs_first"], n_ops=len(rec.tags), sum_of_ops_ms=sum(res[n]["per_op"]), by_shape=by_shape,
                      wall_ms_median=statistics.median(res[n]["wall_ms"]), wall_ms_min=min(res[n]["wall_ms"]), wall_ms=res[n]["wall_ms"],
                      by_descr={k: dict(n=d["n"], ms=round(d["ms"], 4), us_each=round(1e3 * d["ms"] / d["n"], 2),
                                        tflops=round(d["gflop"] / d["ms"], 1) if d["ms"] > 0 else 0) for k, d in
                                sorted(by.items(), key=lambda kv: -kv[1]["ms"])})
        print(f"== {n}: wall median {out[n]['wall_ms_median']:.3f} ms (min {out[n]['wall_ms_min']:.3f}), sum of ops {out[n]['sum_of_ops_ms']:.3f} ms, "
              f"{len(rec.tags)} ops, rel-rms vs {names[0]} {out[n]['rel_rms_vs_first']:.2e}", file=sys.stderr, flush=True)
        for k, d in list(out[n]["by_descr"].items())[:22]:
            print(f"     {k:24s} n={d['n']:4d} {d['ms']:8.3f} ms  {d['us_each']:8.2f} us each  {d['tflops']:8.1f} TF/s", file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
