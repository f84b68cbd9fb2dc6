"""GPU tool: A/B of whole-forward configurations in ONE process on one box (guide rule 24: perf deltas come from interleaved
rounds inside one process).  Records the 1024^2 CFG-2 SDXL forward once per configuration, then times the recorded plans
in interleaved rounds: per-op HIP-event times (min over rounds, summed by op family) and the back-to-back wall time of the
plan (median over rounds).  Knobs per configuration:
    ln_stats   unet.LN_STATS_HANDOVER (LayerNorm statistics handed over from the producing GEMM's epilogue)
    gn_stats   unet.GN_STATS_HANDOVER (GroupNorm statistics likewise)
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

CONFIGS = collections.OrderedDict([
    ("r02", dict(ln_stats=False, xattn=1)),
    ("stats", dict(ln_stats=True, xattn=1)),
    ("stats_x2", dict(ln_stats=True, xattn=2)),
    ("stats_x3", dict(ln_stats=True, xattn=3)),
    ("stats_x4", dict(ln_stats=True, xattn=4)),
    ("stats_x3_geglu128", dict(ln_stats=True, xattn=3, tuning={"2048,10240,1280,0,1": [128, 128, 1], "8192,5120,640,0,1": [128, 128, 1]})),
    ("stats_x3_lin640_24128", dict(ln_stats=True, xattn=3, tuning={"8192,640,640,0": [24128, 160, 1]})),
    # round-3 session C: attention key loop (attn: imh_debug_set key 4) and the resident-key-tile cross-attention
    ("x1_a1", dict(xattn=1, attn=1)),
    ("x3_a1", dict(xattn=3, attn=1)),
    ("x4_a1", dict(xattn=4, attn=1)),
    ("x7_a1", dict(xattn=7, attn=1)),
    ("x3_a2", dict(xattn=3, attn=2)),
    ("x4_a2", dict(xattn=4, attn=2)),
    ("x4_a2_lin640", dict(xattn=4, attn=2, tuning={"8192,640,640,0": [24128, 160, 1]})),
    # session D: wave-specialised projection pair of self-attention (dual_ws), in-loop vs handed-over statistics for the
    # fused cross-attention (xstats)
    ("d_base", dict(xattn=1, attn=1, dual_ws=False, xstats=True)),
    ("d_xloop", dict(xattn=1, attn=1, dual_ws=False, xstats=False)),
    ("d_dualws", dict(xattn=1, attn=1, dual_ws=True, xstats=False)),
    ("d_dualws_pipe", dict(xattn=1, attn=2, dual_ws=True, xstats=False)),
    ("d_all_lin640", dict(xattn=1, attn=2, dual_ws=True, xstats=False, tuning={"8192,640,640,0": [24128, 160, 1]})),
    # session H: persistent two-tile GEGLU kernel (33256), weight rings of the LDS-halo conv (7328 / 7428 x 160)
    ("h_base", dict()),
    ("h_conv64_s3", dict(tuning={f"8192,640,{k},1": [7328, 160, 1] for k in (2880, 5760, 8640, 11520, 17280)})),
    ("h_conv64_s4", dict(tuning={f"8192,640,{k},1": [7428, 160, 1] for k in (2880, 5760, 8640, 11520, 17280)})),
    ("h_conv128_s3", dict(tuning={f"32768,320,{k},1": [7328, 160, 1] for k in (2880, 5760, 8640)})),
    ("h_conv128_s4", dict(tuning={f"32768,320,{k},1": [7428, 160, 1] for k in (2880, 5760, 8640)})),
    ("i_conv128_p16", dict(tuning={f"32768,320,{k},1": [7256, 160, 1] for k in (2880, 5760, 8640)})),
    ("i_conv128_p16_s3", dict(tuning={f"32768,320,{k},1": [7356, 160, 1] for k in (2880, 5760, 8640)})),
    # session J: GroupNorm statistics from the producing conv / GEMM epilogue (gn_stats)
    ("j_gn_off", dict(gn_stats=False)),
    ("j_gn_on", dict(gn_stats=True)),
    # session K (--stacked 4): the S = 4 table; GEGLU with handed-over statistics on 128 x 128 instead of 256 x 160 ws
    # session L: 128 x 160 wave-specialised tiles at TWO workgroups per CU (22128: two-slot rings, 4 consumer + 2 producer waves)
    ("l_base", dict()),
    ("l_geglu", dict(tuning={"2048,10240,1280,0,1": [22128, 160, 1], "8192,5120,640,0,1": [22128, 160, 1]})),
    ("l_geglu32", dict(tuning={"2048,10240,1280,0,1": [22128, 160, 1]})),
    ("l_all", dict(tuning={"2048,10240,1280,0,1": [22128, 160, 1], "8192,5120,640,0,1": [22128, 160, 1], "8192,640,640,0": [22128, 160, 1],
                           "8192,640,2560,0": [22128, 160, 1]})),
    # session M: XCD partition forced for every GEMM (imh_debug_set key 2: 2..5 = (8,1) (4,2) (2,4) (1,8); 0 = cost model)
    ("m_auto", dict()),
    ("m_xcd81", dict(xcd=2)),
    ("m_xcd42", dict(xcd=3)),
    ("m_xcd24", dict(xcd=4)),
    ("m_xcd18", dict(xcd=5)),
    # session N: deferred running maximum in the pipelined attention key loop (attn = 3, the default) vs the textbook rule (2)
    ("n_attn2", dict(attn=2)),
    ("n_attn3", dict(attn=3)),
    ("k_s4", dict()),
    ("k_s4_geglu64_128", dict(tuning={"32768,5120,640,0,1": [128, 128, 1]})),
    ("k_s4_geglu_both128", dict(tuning={"32768,5120,640,0,1": [128, 128, 1], "8192,10240,1280,0,1": [128, 128, 1]})),
])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--configs", default="j_gn_off,j_gn_on")
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
        U.XATTN_STATS_HANDOVER = bool(c.get("xstats", False))
        U.GN_STATS_HANDOVER = bool(c.get("gn_stats", True))
        AP.DUAL_WS = bool(c.get("dual_ws", False))
        lib.imh_debug_set(3, int(c.get("xattn", 0)))
        lib.imh_debug_set(4, int(c.get("attn", 0)))
        lib.imh_debug_set(2, int(c.get("xcd", 0)))
        tun = dict(_load_tuning())
        for k, v in (c.get("tuning") or {}).items():
            tun[tuple(int(x) for x in k.split(","))] = tuple(v)
        rec, out, st = record(u, dtype, 128, S=a.stacked, tuning=tun)
        rec.run()
        torch.cuda.synchronize()
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
    out = {}
    for n in names:
        rec, c = plans[n]
        by = collections.OrderedDict()
        for (tag, kind, descr, fl, by_, *rest), t in zip(rec.tags, res[n]["per_op"]):
            d = by.setdefault(descr, dict(n=0, ms=0.0, gflop=0.0))
            d["n"] += 1; d["ms"] += t; d["gflop"] += fl / 1e9
        out[n] = dict(cfg=c, rel_rms_vs_first=res[n]["rel_rms_vs_first"], n_ops=len(rec.tags), sum_of_ops_ms=sum(res[n]["per_op"]),
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
