"""GPU tool: (1) records one full SDXL UNet forward (1024^2, CFG batch 2) and times every op with
HIP events; (2) sweeps tile / split-K configurations for every distinct GEMM / conv shape of that
forward and writes the winners to tuning.json; (3) re-times the forward with the tuned table.
Outputs JSON under gpurun_out/.  Usage: python tools/sweep.py [--lat 128] [--dtype bf16] [--no-sweep]"""
import argparse
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagharmony_amd import lib as L                                   # noqa: E402
from imagharmony_amd.ctx import Ctx                                    # noqa: E402
from imagharmony_amd.ip_adapter import install_ip_processors           # noqa: E402
from imagharmony_amd.unet import StepState, UNet2DConditionModel, UNetConfig   # noqa: E402

DEV = "cuda:0"


def build_unet(dtype, num_tokens=4):
    t0 = time.time()
    with torch.device(DEV):
        u = UNet2DConditionModel(UNetConfig())
    u.init_random_(1234)
    u = u.to(dtype)
    procs = install_ip_processors(u, num_tokens=num_tokens, device=DEV, dtype=dtype, init="empty")
    g = torch.Generator(device=DEV).manual_seed(99)
    for p in procs.values():
        for q in p.parameters():
            q.data.copy_(torch.randn(q.shape, generator=g, device=DEV) * (q.shape[1] ** -0.5))
    torch.cuda.synchronize()
    print(f"unet built in {time.time() - t0:.1f}s, {torch.cuda.memory_allocated() / 2**30:.2f} GiB", flush=True)
    return u


def record(u, dtype, lat, S=1, T=4, tuning=None, cells=0):
    pre = Ctx(DEV, dtype)
    g = torch.Generator(device="cpu").manual_seed(7)
    ehs = torch.randn(2 * S, 77 + T, 2048, generator=g).to(DEV, dtype)
    pooled = torch.randn(2 * S, 1280, generator=g).to(DEV, dtype)
    ids = torch.tensor([[lat * 8, lat * 8, 0, 0, lat * 8, lat * 8]], dtype=torch.float32).repeat(2 * S, 1).to(DEV)
    st = u.prepare_conditioning(pre, ehs, pooled, ids)
    st.t_value = torch.full((2 * S,), 500.0, device=DEV)
    st.latents = torch.randn(S, 4, lat, lat, generator=g).to(DEV)
    rec = Ctx(DEV, dtype, record=True)
    if tuning is not None:
        rec.tuning = tuning
    rec.xcd_cells = cells
    out = u.emit_forward(rec, st, S, lat, lat, cfg_dup=True)
    return rec, out, st


def time_plan(rec, reps=3):
    rec.run()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        ms = rec.time_ops()
        best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
    # wall time of the whole plan, back to back
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        rec.run()
    e1.record()
    torch.cuda.synchronize()
    return best, e0.elapsed_time(e1) / 3


def summarize(rec, ms, wall_ms, label):
    by = collections.OrderedDict()
    for (tag, kind, descr, fl, by_, *rest), t in zip(rec.tags, ms):
        d = by.setdefault(descr, dict(n=0, ms=0.0, gflop=0.0, mb=0.0))
        d["n"] += 1; d["ms"] += t; d["gflop"] += fl / 1e9; d["mb"] += by_ / 1e6
    tot = sum(ms)
    print(f"== {label}: {len(ms)} ops, sum-of-ops {tot:.2f} ms, back-to-back wall {wall_ms:.2f} ms", flush=True)
    for k, d in sorted(by.items(), key=lambda kv: -kv[1]["ms"]):
        tf = d["gflop"] / d["ms"] if d["ms"] > 0 else 0
        gbs = d["mb"] / d["ms"] if d["ms"] > 0 else 0
        print(f"  {k:22s} n={d['n']:4d} {d['ms']:8.3f} ms  {tf:8.1f} TF/s  {gbs:8.1f} GB/s(alg)")
    return dict(label=label, n_ops=len(ms), sum_ms=tot, wall_ms=wall_ms, by_descr=by)


def sweep_shapes(rec, dtype, extra=None, base=None):
    l = rec.lib
    shapes = collections.OrderedDict()
    for (tag, kind, descr, fl, by_, shape, *rest) in rec.tags:
        if kind == L.OP_GEMM and shape is not None:
            shapes.setdefault(shape, descr)
    print(f"{len(shapes)} distinct GEMM/conv shapes", flush=True)
    ctx = Ctx(DEV, dtype)
    table = {}
    report = []
    for (M, N, K, conv, geom), descr in shapes.items():
        x = w = None
        if conv:
            B, H, W, Cin, stride, up = geom
            x = torch.randn(B, H, W, Cin, device=DEV).to(dtype)
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
            call = lambda cfg, c=None: (c or ctx).conv3x3(x, w, stride=stride, up=up, cfg=cfg, out=out)
            Ho, Wo = ((H << up) - 1) // stride + 1, ((W << up) - 1) // stride + 1
            out = torch.empty(B, Ho, Wo, N, device=DEV, dtype=dtype)
        else:
            x = torch.randn(M, K, device=DEV).to(dtype)
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(dtype)
            out = torch.empty(M, N, device=DEV, dtype=dtype)
            call = lambda cfg, c=None: (c or ctx).gemm(x, w, cfg=cfg, out=out)
        nkt = K // 64
        res = []
        cands = []
        for bm in (128, 64):
            for bn in (128, 64):
                for sp in (1, 2, 4, 8):
                    if sp > 1 and (nkt // sp < 4 or -(-M // bm) * -(-N // bn) * sp > 2048):
                        continue
                    cands.append((bm, bn, sp))
        cands += [(256, 128, 1), (256, 256, 1), (3128, 128, 1), (3128, 128, 2), (3064, 64, 1),
                  (4128, 64, 1), (5064, 64, 1), (4064, 128, 1), (6128, 320, 1), (5258, 320, 1),
                  (1464, 160, 1), (2464, 160, 1), (24128, 160, 1), (24128, 128, 1), (23256, 160, 1)]      # wave-specialised
        if not conv:
            cands += [(8256, 256, 1), (9128, 320, 1), (9256, 320, 1)]
        elif geom[4] == 1:
            cands += [(7128, 320, 1), (7128, 160, 1), (7564, 320, 1), (7564, 160, 1), (7328, 160, 1), (7428, 160, 1), (7256, 160, 1), (7356, 160, 1)]           # LDS-halo conv kernel (stride 1 only)
        if extra:                      # incremental: the current table entry against the new variants only
            key = f"{M},{N},{K},{conv}"
            cands = [tuple(base[key])] if base and key in base else [tuple(ctx._config(M, N, K, conv, 0))]
            cands += [c for c in extra if c not in cands]
        from tools.gemm_bench import graph_time
        for cfg in cands:
            try:
                res.append((graph_time(lambda c: call(cfg, c), dtype, n=10, reps=2), cfg))
            except Exception as ex:          # noqa: BLE001
                print("  cfg failed", cfg, ex)
        res.sort()
        best_ms, best = res[0]
        hb = ctx._config(M, N, K, conv, 0)
        h_ms = [r[0] for r in res if tuple(r[1]) == tuple(hb)]
        tf = 2.0 * M * N * K / best_ms / 1e9
        print(f"  {descr:18s} M={M:6d} N={N:6d} K={K:6d} conv={conv} best {best} {best_ms*1e3:8.1f} us {tf:7.1f} TF/s"
              f" | heuristic {tuple(hb)} {h_ms[0]*1e3 if h_ms else -1:8.1f} us", flush=True)
        table[f"{M},{N},{K},{conv}"] = list(best)
        report.append(dict(descr=descr, M=M, N=N, K=K, conv=conv, best=list(best), best_us=best_ms * 1e3, tflops=tf,
                           all=[(r[0] * 1e3, list(r[1])) for r in res]))
    return table, report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lat", type=int, default=128)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--candidates", type=int, default=1, help="PNS candidates stacked per forward (UNet batch 2S)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--extra", default="", help="incremental sweep: 'bm,bn,splits;...' tried against the current tuning.json entry only")
    a = ap.parse_args()
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
    os.makedirs(a.out, exist_ok=True)
    print(torch.cuda.get_device_name(0), flush=True)
    u = build_unet(dtype)
    rec, out, st = record(u, dtype, a.lat, S=a.candidates)
    ms, wall = time_plan(rec)
    results = [summarize(rec, ms, wall, "heuristic configs")]
    assert torch.isfinite(out.float()).all(), "non-finite UNet output"
    if not a.no_sweep:
        extra = [tuple(int(v) for v in c.split(",")) for c in a.extra.split(";") if c]
        base = json.load(open(os.path.join(ROOT, "imagharmony_amd", "tuning.json"))) if extra else None
        table, report = sweep_shapes(rec, dtype, extra, base)
        with open(os.path.join(a.out, f"tuning_s{a.candidates}.json" if a.candidates > 1 else "tuning.json"), "w") as f:
            json.dump(table, f, indent=0)
        with open(os.path.join(a.out, "sweep_report.json"), "w") as f:
            json.dump(report, f)
        tuned = {tuple(int(v) for v in k.split(",")): tuple(v) for k, v in table.items()}
        rec2, out2, _ = record(u, dtype, a.lat, S=a.candidates, tuning=tuned)
        ms2, wall2 = time_plan(rec2)
        results.append(summarize(rec2, ms2, wall2, "tuned configs"))
        rec2.capture()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rec2.replay(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            rec2.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"hipGraph replay: {e0.elapsed_time(e1) / 5:.2f} ms per UNet forward", flush=True)
        results.append(dict(label="graph replay", wall_ms=e0.elapsed_time(e1) / 5))
    with open(os.path.join(a.out, "forward_profile.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
