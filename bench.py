#!/usr/bin/env python
"""Headline benchmark: 1024^2 SDXL images/sec (30 DDIM steps, one PNS candidate seed per GPU per step).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" = one full 30-step denoise of one candidate per rank through the HIP hot path (UNet forward with
the harmony-aware IP-Adapter processors + CFG + DDIM update), at BASELINE.json configs[1] (N=1) /
configs[2] (N=8): SDXL 1024x1024, batch 1 x CFG 2, IP scale 1.0, 4 image tokens, bf16.  Inputs are
synthetic (seeded random weights of the exact SDXL architecture, random text / image embeddings,
SURVEY.md 8d) and resident in HBM before the timed region.  Candidates are independent, so ranks share
nothing per step (weak scaling); RCCL is used for the one-time weight / conditioning broadcast and the
final score gather only.  The K/V projections of the conditioning are computed once per PNS run (they are
step- and seed-invariant), outside the timed region, as the one-time conditioning work they are.

Rank 0 prints ONE JSON line with the metric plus
  roofline     -- the dominant kernel family (MFMA GEMM / implicit-GEMM conv): algorithmic FLOPs of its
                  launches in one denoise step / their summed durations, measured here with HIP events on
                  the launch stream; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).
  cpu_baseline -- the CPU oracle (reference processors restated + restated diffusers UNet, fp32) timed on
                  this box's host cores for ONE 1024^2 CFG-2 UNet forward, extrapolated x30 steps.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--denoise-steps", type=int, default=30)
    ap.add_argument("--ip-tokens", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-res", type=int, default=1024)
    ap.add_argument("--in-flight", type=int, default=2, help="EXTRA measurement (never `value`): PNS candidates in flight per "
                    "GPU, one HIP stream each, batch 1 each; 1 = skip")
    ap.add_argument("--stacked", type=int, default=4, help="EXTRA measurement (never `value`): PNS candidates stacked into one "
                    "UNet batch per GPU (BASELINE.json configs[4] runs 4 per GPU); 1 = skip")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm; gloo only for "
                    "single-GPU testing of the multi-rank control path)")
    return ap.parse_args()


def build_unet(device, dtype, ip_tokens):
    from imagharmony_amd.ip_adapter import install_ip_processors
    from imagharmony_amd.unet import UNet2DConditionModel, UNetConfig
    with torch.device(device):
        u = UNet2DConditionModel(UNetConfig())
    u.init_random_(1234)
    u = u.to(dtype)
    procs = install_ip_processors(u, num_tokens=ip_tokens, scale=1.0, device=device, dtype=dtype, init="empty")
    g = torch.Generator(device=device).manual_seed(4321)
    for p in procs.values():
        for q in p.parameters():
            q.data.copy_(torch.randn(q.shape, generator=g, device=device) * (q.shape[1] ** -0.5))
    return u


def synthetic_conditioning(T_ip):
    """prompt / negative embeds N(0,1) (seeds 1000/1001), IP tokens (seed 2000)."""
    g = lambda s: torch.Generator("cpu").manual_seed(s)
    pe = torch.cat([torch.randn(1, 77, 2048, generator=g(1000)), torch.randn(1, T_ip, 2048, generator=g(2000))], 1)
    ne = torch.cat([torch.randn(1, 77, 2048, generator=g(1001)), torch.randn(1, T_ip, 2048, generator=g(2001))], 1)
    po, no = torch.randn(1, 1280, generator=g(1002)), torch.randn(1, 1280, generator=g(1003))
    return pe, ne, po, no


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(res, ip_tokens, denoise_steps):
    """the fp32 CPU oracle on this box's host cores -- 1 warm-up + 2 timed UNet forwards at
    res^2 (CFG batch 2), extrapolated x denoise_steps; plus BASELINE.json configs[0] (512^2, 10 DDIM steps, the
    designated CPU config) from 2 timed forwards.  `kind` is "port": the oracle restates the reference's processors
    and the absent diffusers UNet; /root/reference itself does not exist on the GPU box."""
    from oracle.pipeline import install_ip_processors
    from oracle.sdxl_unet import UNet2DConditionModel, sdxl_config
    # 32 threads: measured best on the 2x64-core EPYC host of the MI355X box (128 / 256 threads are 4-5x slower)
    nthreads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    t0 = time.time()
    with torch.device("meta"):
        u = UNet2DConditionModel(sdxl_config())
        install_ip_processors(u, num_tokens=ip_tokens)
    u = u.to_empty(device="cpu").eval()
    with torch.no_grad():        # cheap seeded fill: one random block tiled into every weight (timing is value-independent)
        block = torch.randn(1 << 22, generator=torch.Generator().manual_seed(1))
        for p in u.parameters():
            if p.ndim >= 2:
                flat, sc = p.view(-1), p[0].numel() ** -0.5
                for i in range(0, flat.numel(), block.numel()):
                    n = min(block.numel(), flat.numel() - i)
                    torch.mul(block[:n], sc, out=flat[i:i + n])
            else:
                p.fill_(1.0)
    build_s = time.time() - t0

    def fwd_times(r, n_warm, n_timed):
        lat = r // 8
        x = torch.randn(2, 4, lat, lat)
        ehs = torch.randn(2, 77 + ip_tokens, 2048)
        kw = {"text_embeds": torch.randn(2, 1280), "time_ids": torch.tensor([[r, r, 0, 0, r, r]] * 2, dtype=torch.float32)}
        ts = []
        with torch.no_grad():
            for i in range(n_warm + n_timed):
                t1 = time.time()
                u(x, torch.tensor(500.0), ehs, added_cond_kwargs=kw)
                if i >= n_warm:
                    ts.append(time.time() - t1)
        return ts

    quick = os.environ.get("IMH_BENCH_CPU_QUICK") == "1"          # 1 un-warmed forward only (profiling runs)
    ts = fwd_times(res, 0 if quick else 1, 1 if quick else 2)
    fwd_s = sum(ts) / len(ts)
    out = {"value": 1.0 / (fwd_s * denoise_steps), "unit": "images/sec", "cores": nthreads, "kind": "port",
           "cpu_model": _cpu_model(), "host_cores_total": os.cpu_count(),
           "sample": f"{len(ts)} timed UNet forward(s) of the fp32 CPU oracle at {res}x{res}, CFG batch 2, after "
                     f"{0 if quick else 1} warm-up ({', '.join(f'{t:.1f}' for t in ts)} s; model build {build_s:.0f} s not "
                     f"counted), extrapolated x{denoise_steps} DDIM steps per image",
           "seconds_per_unet_forward": fwd_s}
    if not quick:
        t5 = fwd_times(512, 0, 2)
        f5 = sum(t5) / len(t5)
        out["configs0_512x512_10_steps"] = {"value": 1.0 / (f5 * 10), "unit": "images/sec", "seconds_per_image": f5 * 10,
                                            "sample": f"2 timed forwards at 512x512 ({', '.join(f'{t:.1f}' for t in t5)} s), x10 DDIM steps"}
    return out


def configs3_extra(device, res_px, S=4, T=16, steps=50, dtype=torch.float16, n_embeds=1, note=None):
    """BASELINE.json configs[3]: SDXL 1024^2, 50 steps, batch 4 per GPU (S candidates stacked into one UNet batch of 2S with
    CFG), Resampler num_queries = 16 (the 16 image tokens come out of the IP-Adapter-Plus-XL Resampler on the device:
    ip_adapter.py:392-403), fp16.  Own UNet (random weights) + engine; one warm-up denoise, one timed.
    configs[4] is the same harness at bf16, 30 steps, T = 32 = two 16-token image embeds (dual-category edit, n_embeds = 2)."""
    from imagharmony_amd import pns
    from imagharmony_amd.modules import Resampler
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    from imagharmony_amd.schedulers import DDIMScheduler
    unet = build_unet(device, dtype, T)
    q = T // n_embeds
    rs = Resampler(dim=1280, depth=4, dim_head=64, heads=20, num_queries=q, embedding_dim=1280, output_dim=2048, ff_mult=4).to(device, dtype).eval()
    with torch.no_grad():                                 # [2, T, 2048]: image tokens of the positive / negative branch, one Resampler
        ip = torch.cat([rs(torch.randn(2, 257, 1280, generator=torch.Generator("cpu").manual_seed(5 + j)).to(device, dtype)).float().cpu()
                        for j in range(n_embeds)], 1)     # call per image embed (ip_adapter.py:405-417), tokens concatenated
    pe, ne, po, no = synthetic_conditioning(T)
    pe[:, 77:], ne[:, 77:] = ip[:1], ip[1:]
    pe, ne, po, no = [t.to(device) for t in (pe, ne, po, no)]
    pipe = StableDiffusionXLCustomPipeline(unet, scheduler=DDIMScheduler(), device=device, dtype=dtype)
    e = pipe.engine.__class__(unet, device, dtype, True)
    e.set_conditioning(pe.repeat(S, 1, 1), ne.repeat(S, 1, 1), po.repeat(S, 1), no.repeat(S, 1), res_px, res_px, guidance_scale=5.0)
    e.set_schedule(pipe.scheduler, steps)
    z = torch.cat([pns.seed_latents(3000 + j, (1, 4, res_px // 8, res_px // 8)) for j in range(S)], 0).to(device)
    e.denoise(z)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    o = e.denoise(z)
    torch.cuda.synchronize(device)
    dt_s = time.perf_counter() - t0
    out = {"images_per_sec": S / dt_s, "ms_per_unet_forward": dt_s / steps * 1e3, "unet_batch": 2 * S, "denoise_steps": steps,
           "ip_tokens": T, "dtype": "fp16" if dtype == torch.float16 else "bf16", "scheduler": "DDIM", "outputs_finite": bool(torch.isfinite(o).all().item()),
           "note": note or "BASELINE.json configs[3] on one GPU (never `value`): the 16 image tokens are produced by the Resampler on the "
                   "device; parity of this shape family: tests/test_gpu_parity_fullsize.py (UNet batch 8, T = 16, fp16 vs the CPU oracle)"}
    del e, pipe, unet, rs
    torch.cuda.empty_cache()
    return out


def resampler_extra(device, dtype):
    """ms per call of the IP-Adapter-Plus-XL Resampler (ip_adapter/resampler.py:81-158 at the ip_adapter.py:392-403 config:
    dim 1280, depth 4, 20 heads x 64, 16 queries, 257 CLIP patch tokens -> [B, 16, 2048]) on the HIP kernels, eager and
    replayed from a captured graph, against the weight-read bound (every weight byte once at 6.3 TB/s)."""
    from imagharmony_amd.modules import Resampler
    m = Resampler(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, embedding_dim=1280, output_dim=2048, ff_mult=4)
    m = m.to(device, dtype).eval()
    wbytes = sum(p.numel() * p.element_size() for p in m.parameters())
    out = {"weight_bytes": wbytes, "weight_read_bound_us": wbytes / 6.3e12 * 1e6, "ms_per_call": {}}
    for B in (1, 4):
        x = torch.randn(B, 257, 1280, generator=torch.Generator("cpu").manual_seed(5)).to(device, dtype)
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(10):
            y = m(x)
        torch.cuda.synchronize(device)
        eager = (time.perf_counter() - t0) / 10 * 1e3
        s = torch.cuda.Stream(device)
        s.wait_stream(torch.cuda.current_stream(device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            m(x)
            torch.cuda.synchronize(device)
            with torch.cuda.graph(g, stream=s):
                yg = m(x)
        torch.cuda.current_stream(device).wait_stream(s)
        g.replay(); torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize(device)
        row = {"eager": eager, "graph_replay": (time.perf_counter() - t0) / 20 * 1e3,
               "outputs_finite": bool(torch.isfinite(yg.float()).all().item()), "replay_equals_eager": bool(torch.equal(yg, y))}
        # what IPAdapterPlusXL.get_image_embeds runs since round 5 (imagharmony_amd.modules.Graphed: copy in, replay, clone out)
        from imagharmony_amd.modules import Graphed
        gm = Graphed(m)
        ya = gm(x)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(20):
            ya = gm(x)
        torch.cuda.synchronize(device)
        row["as_get_image_embeds_runs_it"] = (time.perf_counter() - t0) / 20 * 1e3
        row["adapter_path_equals_eager"] = bool(torch.equal(ya, y))
        out["ms_per_call"][f"B={B}"] = row
    out["note"] = ("once per image (twice with the unconditional branch, ip_adapter.py:413-416); ~45 launches of 1-16-row GEMMs, "
                   "LayerNorms and the LDS-resident 16 x 273 latent cross-attention: launch-latency-bound, not on the per-step path")
    return out


# tile (rows of the token operand, rows of the weight operand) staged per K tile by each variant code (include/imh.h)
_TILE = {1464: (64, None), 2464: (64, None), 24128: (128, None), 23256: (256, None), 9128: (128, 320), 9256: (256, 320), 8256: (256, 256),
         5258: (256, 320), 5256: (256, 320), 6128: (128, 320), 6064: (64, 160), 7064: (64, 160), 256: (256, None),
         3128: (128, 128), 3064: (64, 64), 4064: (64, None), 4128: (128, 64), 5064: (64, 64)}


def lds_operand_bytes(shape, epi, esz=2):
    """bytes a GEMM / conv launch moves over the L2 -> LDS path (LDS-DMA): every workgroup stages its token tile and its weight
    tile once per 64-wide K tile -- tiles x (BM + BN) x K x 2 B; the LDS-halo conv stages the (PH + 2) x 18 pixel halo once per
    64-channel chunk and the weight tile once per (chunk, tap).  This, not HBM bytes or FLOPs, is what the M = 2048 layers of
    the forward are bound by."""
    if shape is None or epi is None or "cfg" not in epi:
        return 0.0
    M, N, K, conv, geom = shape
    bm, bn, sp = epi["cfg"]
    if bm in (7128, 7564, 7328, 7428, 7256, 7356):
        B, H, W, Cin, stride, up = geom
        ph = 4 if bm == 7564 else (16 if bm in (7256, 7356) else 8)
        Ho, Wo = H << up, W << up
        tiles = B * -(-Ho // ph) * -(-Wo // 16) * -(-N // bn)
        return float(tiles) * (Cin // 64) * ((ph + 2) * 18 * 128 + 9 * bn * 128)
    tm, tn = _TILE.get(bm, (bm, bn))
    tn = tn or bn
    return float(-(-M // tm)) * -(-N // tn) * (tm + tn) * K * esz


def ip_attn_cfg4(device, dtype, reps=20):
    """the north-star call at BASELINE.json configs[3] / SURVEY 8a 'cfg4' (UNet batch 8, 16 Resampler tokens): one
    IPAttnProcessor2_0 call on an IP-active layer = fused [to_q + norm2 + text SDPA + image-prompt SDPA + axpy] launch + [to_out +
    residual] launch, `reps` calls recorded into a plan and timed with HIP events on the launch stream"""
    from imagharmony_amd import lib as L
    from imagharmony_amd.attention_processor import IPAttnProcessor2_0, fold_ln
    from imagharmony_amd.ctx import Ctx
    from imagharmony_amd.unet import Attention, Norm
    B, Lq, C_, T, H = 8, 1024, 1280, 16, 20
    g = torch.Generator(device="cpu").manual_seed(11)
    attn = Attention(C_, H, cross_attention_dim=2048)
    proc = IPAttnProcessor2_0(C_, 2048, scale=1.0, num_tokens=T)
    norm = Norm(C_, 1e-5)
    with torch.no_grad():
        for p in list(attn.parameters()) + list(proc.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (p.shape[-1] ** -0.5 if p.ndim > 1 else 0.02))
        norm.weight.fill_(1.0); norm.bias.zero_()
    attn, proc, norm = attn.to(device, dtype), proc.to(device, dtype), norm.to(device, dtype)
    x = torch.randn(B * Lq, C_, generator=g).to(device, dtype)
    ehs = torch.randn(B, 77 + T, 2048, generator=g).to(device, dtype)
    pre = Ctx(device, dtype)
    kv = proc.prepare_kv(pre, attn, ehs)
    rec = Ctx(device, dtype, record=True)
    for _ in range(reps):
        y = proc.emit(rec, attn, x, B, Lq, residual=x, kv=kv, ln=norm)
        rec.free(y)
    fl = sum(t[3] for t in rec.tags) / reps
    rec.run(); torch.cuda.synchronize(device)
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rec.run(); e1.record(); torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / reps
        best = us if best is None else min(best, us)
    kv_fl = 2.0 * 2 * B * (77 + T) * 2048 * C_
    return {"bound": "mfma", "achieved": fl / (best * 1e-6) / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": fl / (best * 1e-6) / 1e12 / 2500.0,
            "avg_call_us": best, "algorithmic_gflop_per_call": fl / 1e9, "algorithmic_gflop_per_call_survey_8a": (fl + kv_fl) / 1e9,
            "shape": f"UNet batch {B} (4 candidates x CFG), L = {Lq}, C = {C_}, {T} image tokens",
            "note": f"{reps} back-to-back calls (warm operands), HIP events around the recorded plan; K/V projections of the conditioning "
                    "are step-invariant and not in the timed call"}


def box_calibration(device, dtype, reps=20):
    """one FIXED launch timed on this box, so that numbers from different boxes / rounds can be normalised (the same build measures
    20.5-23.1 ms per forward from box to box): the 8192 x 5120 x 2560 GEMM on the wave-specialised 256 x 160 kernel (variant 23256),
    `reps` back-to-back launches from a recorded plan between two HIP events"""
    from imagharmony_amd.ctx import Ctx
    M, N, K = 8192, 5120, 2560
    g = torch.Generator(device="cpu").manual_seed(77)
    x = torch.randn(M, K, generator=g).to(device, dtype)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(device, dtype)
    y = torch.empty(M, N, device=device, dtype=dtype)
    rec = Ctx(device, dtype, record=True)
    for _ in range(reps):
        rec.gemm(x, w, out=y, cfg=(23256, 160, 1), descr="calibration")
    rec.run(); torch.cuda.synchronize(device)
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rec.run(); e1.record(); torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / reps
        best = us if best is None else min(best, us)
    fl = 2.0 * M * N * K
    # ... and the clock / power the part sustains under this launch alone (~2 s of back-to-back replays, sampled from the host): a dense MFMA
    # kernel is power-limited well below 2.4 GHz, so its fraction of the peak AT THAT CLOCK is what says how busy the matrix pipes are
    clk = ClockSampler(period=0.25)
    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        rec.run()
        torch.cuda.synchronize(device)
    clocks = clk.stop()
    tf = fl / (best * 1e-6) / 1e12
    return {"launch": "imh::gemm_ws_kernel 256x160 (variant 23256), 8192 x 5120 x 2560, bf16 random operands", "us": best,
            "tflops": tf, "reference_us": 186.0, "clocks_under_this_launch": clocks,
            "frac_of_peak_at_sampled_clock": (tf / clocks["mfma_peak_at_this_clock_tflops"]) if clocks else None,
            "note": "reference_us = the same launch on the round-3 profile box (profiles/r03_pmc_sq_gemm_attn.md); us / reference_us "
                    "scales this box against it"}


class ClockSampler:
    """shader clock and socket power while the timed steps run, sampled from the host (rocm-smi reads sysfs; nothing is launched on the
    GPU): the part sustains 1.8-2.1 GHz under these kernels, not the 2.4 GHz its MFMA peak is quoted at (profiles/r04_clocks_under_load.txt)"""
    def __init__(self, period=1.0):
        import re, subprocess, threading
        self.rows, self._stop, self._re, self._sp = [], False, re, subprocess
        self.period = period
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        while not self._stop:
            try:
                t = self._sp.run("rocm-smi --showpower --showclocks", shell=True, capture_output=True, text=True, timeout=5).stdout
                m1 = self._re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", t)
                m2 = self._re.search(r"Package Power \(W\): ([\d.]+)", t)
                if m1 and m2:
                    self.rows.append((int(m1.group(1)), float(m2.group(1))))
            except Exception:
                pass
            time.sleep(self.period)

    def stop(self):
        self._stop = True
        self.th.join(timeout=10)
        rows = [r for r in self.rows if r[0] > 500]              # (a sample that caught the idle state between steps is not "under load")
        if not rows:
            return None
        return {"sclk_mhz": sum(r[0] for r in rows) / len(rows), "watts": sum(r[1] for r in rows) / len(rows), "samples": len(rows),
                "mfma_peak_at_this_clock_tflops": 2500.0 * (sum(r[0] for r in rows) / len(rows)) / 2400.0,
                "note": "rocm-smi sampled from the host during the timed steps; `roofline.peak` stays the 2.4 GHz figure"}


def pns_two_stage(eng, pipe, unet, cond, device, dtype, lat_shape, res, N=8, preview_steps=10, final_steps=30, stack=4):
    """the two-stage schedule of assets/1.png / README.md:27 on ONE rank, end to end: N candidate seeds x a `preview_steps` denoise,
    the judge, then the judged-best noise x the full `final_steps` denoise -- so that the serial tail of the scheme is a number
    (at W ranks with N = W the tail runs on the owner rank while the others idle: (N * p + f) / (p + f) is the ceiling of the
    speed-up, 2.75x for 8 x 10 + 30).  Uses the latent-statistic judge (off the measured path's critical work, like bench `value`).
    The previews run `stack` candidates per UNet forward (UNet batch 2 * stack with CFG) -- the mode IPAdapterXL.generate_pns picks by
    itself when a rank holds more than one seed (round 6) -- and, for comparison, one at a time (rounds 3-5); both modes must name
    the same winner."""
    from imagharmony_amd import pns
    pe, ne, po, no = cond
    S = max(1, min(int(stack), N))
    prev1 = eng.fork()
    prev1.set_schedule(pipe.scheduler, preview_steps)
    prevS = eng.__class__(unet, device, dtype, True)
    prevS.set_conditioning(pe.repeat(S, 1, 1), ne.repeat(S, 1, 1), po.repeat(S, 1), no.repeat(S, 1), res, res, guidance_scale=5.0)
    prevS.set_schedule(pipe.scheduler, preview_steps)
    noises = [pns.seed_latents(5000 + j, lat_shape).to(device) for j in range(N)]
    groups = [torch.cat(noises[j:j + S], 0) for j in range(0, N - N % S, S)]
    tail = noises[N - N % S:]                                  # N not a multiple of the stack: the rest one at a time
    prev1.denoise(noises[0]); prevS.denoise(groups[0]); eng.denoise(noises[0])      # plans recorded / captured outside the timed region
    torch.cuda.synchronize(device)

    def run(previews):
        t0 = time.perf_counter()
        scores = previews()
        best = int(torch.argmax(torch.cat(scores)))            # (reads the scores back: the previews are done)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        out = eng.denoise(noises[best])
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        return best, t1 - t0, t2 - t1, out, torch.cat(scores)

    b1, p1, f1, _, sc1 = run(lambda: [pns.default_scorer(prev1.denoise(z)) for z in noises])
    bS, pS, fS, out, scS = run(lambda: [pns.default_scorer(prevS.denoise(g)) for g in groups] +
                                       [pns.default_scorer(prev1.denoise(z)) for z in tail])
    fw = N * preview_steps + final_steps
    tot = pS + fS
    return {"N": N, "preview_steps": preview_steps, "final_steps": final_steps, "previews_per_forward": S,
            "seconds_per_pns_run": tot, "final_images_per_sec": 1.0 / tot, "preview_seconds": pS, "final_seconds": fS,
            "best_seed_index": bS, "unet_forwards_equivalent": fw, "ms_per_unet_forward_equivalent": tot / fw * 1e3,
            "one_at_a_time": {"seconds_per_pns_run": p1 + f1, "preview_seconds": p1, "final_seconds": f1, "best_seed_index": b1},
            "same_winner_in_both_modes": bool(b1 == bS),
            "max_abs_score_difference_between_modes": float((sc1 - scS).abs().max().item()),
            "serial_tail_share_at_8_ranks": f1 / (p1 / N + f1),
            "ceiling_speedup_8_ranks": (p1 + f1) / (p1 / N + f1), "outputs_finite": bool(torch.isfinite(out).all().item()),
            "note": "one rank runs all N previews, then the final denoise of the judged-best noise.  `seconds_per_pns_run` = previews stacked "
                    f"{S} per UNet forward (pns.run_pns(batch={S}), the default of generate_pns when a rank holds several seeds); `one_at_a_time` = "
                    "the rounds 3-5 mode.  At 8 ranks the previews shard 8 ways (one seed per rank, batch 1) and the final denoise stays on the "
                    "owner rank (pns.run_pns) unless its CFG halves are split over two ranks (DenoiseEngine.denoise_cfg_split through "
                    "pns.run_pns(final_split_fn=...): measured no faster, DESIGN.md 7); the ceiling / tail share are for that one-seed-per-rank case"}


def _device_identity(device):
    """something that differs between two physical GPUs of one node and is equal for two ranks on the same one"""
    pr = torch.cuda.get_device_properties(device)
    ident = getattr(pr, "uuid", None)
    if ident is not None:
        return str(ident)
    pci = tuple(getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    return str(pci) if any(v is not None for v in pci) else f"index {device.index}"


def _self_launch(n, script=None, argv=None):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU, RCCL) ourselves through
    torch.distributed.run; rank 0's single JSON line goes to our stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + \
          list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    share_gpu = os.environ.get("IMH_BENCH_SHARE_GPU") == "1"      # testing aid: every rank on GPU 0 (needs --backend gloo)
    if share_gpu:
        local = 0
    # a multi-GPU line must be what it says it is: the launcher's world, the backend's world and --gpus agree, and no two ranks
    # share a device (a mis-launched "8-GPU" run on fewer devices would otherwise report a plausible-looking number)
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N starts them itself)")
    if world > 1 and dist.get_world_size() != a.gpus:
        raise SystemExit(f"bench.py: the {a.backend} backend reports world size {dist.get_world_size()}, --gpus says {a.gpus}")
    if world > 1 and not share_gpu and local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has LOCAL_RANK={local} but only {torch.cuda.device_count()} devices are visible")
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    if world > 1:
        ids = [None] * world
        dist.all_gather_object(ids, _device_identity(device))
        if not share_gpu and len(set(ids)) != world:
            raise SystemExit(f"bench.py: ranks share a device: {ids}")
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]

    from imagharmony_amd import lib as L
    from imagharmony_amd import pns
    from imagharmony_amd.pipeline import StableDiffusionXLCustomPipeline
    from imagharmony_amd.schedulers import DDIMScheduler
    L.load()

    unet = build_unet(device, dtype, a.ip_tokens)
    collectives = None
    if world > 1:
        # the one-time weight broadcast over xGMI (RCCL), timed: first call (communicator set-up + transfer), then once more warm --
        # outside `value`, reported under config.distributed.collectives
        w_bytes = sum(t.numel() * t.element_size() for t in list(unet.parameters()) + list(unet.buffers()))
        bt = []
        for _ in range(2):
            torch.cuda.synchronize(device); dist.barrier(); torch.cuda.synchronize(device)
            tb = time.perf_counter()
            n_coll = pns.broadcast_module_(unet, src=0)
            torch.cuda.synchronize(device); dist.barrier()
            bt.append(time.perf_counter() - tb)
        collectives = {"weight_broadcast": {"bytes": w_bytes, "GB": w_bytes / 1e9, "collectives": n_coll, "first_call_s": bt[0], "warm_s": bt[1],
                                            "warm_GB_per_s": w_bytes / 1e9 / bt[1],
                                            "note": "pns.broadcast_module_: the UNet replica of rank 0 to every rank as flat 512-MB buckets "
                                                    "(dist.broadcast, backend as reported); once per process, never in `value`"}}
    pe, ne, po, no = synthetic_conditioning(a.ip_tokens)
    cond = [t.to(device) for t in (pe, ne, po, no)]
    pns.broadcast_tensors_(cond, src=0)
    pe, ne, po, no = cond
    pipe = StableDiffusionXLCustomPipeline(unet, scheduler=DDIMScheduler(), device=device, dtype=dtype)
    eng = pipe.engine
    # per-image conditioning work (text / image-prompt K, V caches of the 70 cross-attention layers + the added-conditioning embedding:
    # attention_processor.py:410-411,432-433 hoisted out of the step loop) -- outside `value`, reported as conditioning_prepare_ms
    cond_ms = []
    for _ in range(3):
        torch.cuda.synchronize(device)
        tc = time.perf_counter()
        eng.set_conditioning(pe, ne, po, no, a.res, a.res, guidance_scale=5.0)
        torch.cuda.synchronize(device)
        cond_ms.append((time.perf_counter() - tc) * 1e3)
    eng.set_schedule(pipe.scheduler, a.denoise_steps)
    lat_shape = (1, 4, a.res // 8, a.res // 8)
    noises = [pns.seed_latents(i * world + rank, lat_shape).to(device) for i in range(a.warmup + a.steps)]

    def run_concurrent(k, n_steps, n_warm):
        """k candidates in flight on this GPU: k engines (shared weights + conditioning), one stream each"""
        engs = [eng] + [eng.fork() for _ in range(k - 1)]
        streams = [torch.cuda.Stream(device) for _ in range(k)]
        zs = [pns.seed_latents(1000 + j, lat_shape).to(device) for j in range(k)]
        def one_round():
            cur = torch.cuda.current_stream(device)
            for e, s, z in zip(engs, streams, zs):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    e.denoise(z)
            for s in streams:
                cur.wait_stream(s)
        for _ in range(n_warm):
            one_round()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            one_round()
        torch.cuda.synchronize(device)
        return k * n_steps / (time.perf_counter() - t0)

    def run_stacked(S, n_steps, n_warm):
        """S candidates of this rank stacked into one UNet batch (2S with CFG): pns.run_pns(batch=S)"""
        e = eng.__class__(unet, device, dtype, True)
        e.set_conditioning(pe.repeat(S, 1, 1), ne.repeat(S, 1, 1), po.repeat(S, 1), no.repeat(S, 1), a.res, a.res,
                           guidance_scale=5.0)
        e.set_schedule(pipe.scheduler, a.denoise_steps)
        z = torch.cat([pns.seed_latents(2000 + j, lat_shape) for j in range(S)], 0).to(device)
        for _ in range(n_warm):
            e.denoise(z)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            o = e.denoise(z)
        torch.cuda.synchronize(device)
        dt_s = time.perf_counter() - t0
        return S * n_steps / dt_s, bool(torch.isfinite(o).all().item())

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    scores = []
    for i in range(a.warmup):
        eng.denoise(noises[i])
    barrier()
    clk = ClockSampler() if rank == 0 else None              # host thread reading rocm-smi (sysfs): shader clock / power under load
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        out = eng.denoise(noises[i])
        scores.append(pns.default_scorer(out))               # tiny; the PNS judge input
    barrier()
    dt = time.perf_counter() - t0
    clocks = clk.stop() if clk else None
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    per_rank = [dt / a.steps * 1e3]
    if world > 1:
        alld = [torch.empty_like(tmax) for _ in range(world)]
        dist.all_gather(alld, tmax)                          # per-rank wall time: the first real multi-GPU run diagnoses itself
        per_rank = [float(t.item()) / a.steps * 1e3 for t in alld]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        allscores = [torch.empty_like(torch.cat(scores)) for _ in range(world)]
        dist.all_gather(allscores, torch.cat(scores))        # final gather of the candidate scores
        torch.cuda.synchronize(device)
        tg = time.perf_counter()
        dist.all_gather(allscores, torch.cat(scores))        # ... once more, warm and timed (outside `value`)
        torch.cuda.synchronize(device)
        collectives["score_all_gather"] = {"floats_per_rank": int(torch.cat(scores).numel()), "warm_s": time.perf_counter() - tg,
                                           "note": "the only collective after a PNS run's denoises besides the winner broadcast (pns.run_pns)"}
    dt = float(tmax.item())
    finite = bool(torch.isfinite(out).all().item())

    if rank == 0:
        # ---- roofline of the dominant kernel family, timed with HIP events on the launch stream ----
        rec = eng.plan
        eng.eager.ew(L.EW_STEP_SET, eng.st.step, i=(0, 1, 0, 0, 0, 0), descr="step=0")
        ms = rec.time_ops()
        ms = [min(x, y) for x, y in zip(ms, rec.time_ops())]
        g_fl = sum(t[3] for t, m in zip(rec.tags, ms) if t[1] == L.OP_GEMM)
        g_ms = sum(m for t, m in zip(rec.tags, ms) if t[1] == L.OP_GEMM)
        g_by = sum(t[4] for t, m in zip(rec.tags, ms) if t[1] == L.OP_GEMM)
        g_lds = sum(lds_operand_bytes(t[5], t[6]) for t in rec.tags if t[1] == L.OP_GEMM and t[5] is not None)
        g_lds_ms = sum(m for t, m in zip(rec.tags, ms) if t[1] == L.OP_GEMM and t[5] is not None)
        n_g = sum(1 for t in rec.tags if t[1] == L.OP_GEMM)
        tot_fl = sum(t[3] for t in rec.tags)
        achieved = g_fl / (g_ms * 1e-3) / 1e12
        a_fl = sum(t[3] for t, m in zip(rec.tags, ms) if t[1] == L.OP_ATTN)
        a_ms = sum(m for t, m in zip(rec.tags, ms) if t[1] == L.OP_ATTN)
        n_a = sum(1 for t in rec.tags if t[1] == L.OP_ATTN)
        # the north-star kernel: one IPAttnProcessor2_0 call on an IP-active layer = [fused to_q + text attention +
        # image-prompt attention] launch + [to_out] launch; K/V of the conditioning are step-invariant (computed once
        # per image by prepare_conditioning and counted once per image)
        ip_idx = [i for i, t in enumerate(rec.tags) if t[1] == L.OP_XATTN and t[2] == "cross.fused+ip"]
        ip_us, ip_fl = [], []
        for i in ip_idx:
            j = next(k for k in range(i + 1, len(rec.tags)) if rec.tags[k][2] == "cross.to_out")
            ip_us.append((ms[i] + ms[j]) * 1e3)
            ip_fl.append(rec.tags[i][3] + rec.tags[j][3])
        x_fl = sum(t[3] for t, m in zip(rec.tags, ms) if t[1] == L.OP_XATTN)
        x_ms = sum(m for t, m in zip(rec.tags, ms) if t[1] == L.OP_XATTN)
        n_x = sum(1 for t in rec.tags if t[1] == L.OP_XATTN)
        # HBM bytes per launch of the family from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
        # runs of this same command, FETCH_SIZE x2 for gfx950): measured offline, committed under profiles/
        traffic, traffic_src = None, None
        try:
            pj = next(q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json",
                                                                          "r01_pmc_hbm_traffic_final.json")) if os.path.exists(q))
            traffic = json.load(open(pj))["gemm_family"]["hbm_bytes_per_launch"]
            traffic_src = f"profiles/{os.path.basename(pj)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of this command with --denoise-steps 4)"
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        images = a.steps * world
        res = {
            "metric": "1024^2 SDXL images/sec (30 DDIM steps, PNS N seeds)" if a.res == 1024 and a.denoise_steps == 30
                      else f"{a.res}^2 SDXL images/sec ({a.denoise_steps} DDIM steps, PNS N seeds)",
            "value": images / dt, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"SDXL UNet {a.res}x{a.res}, {a.denoise_steps} DDIM steps, CFG 5.0 (UNet batch 2), "
                                   f"IP-Adapter scale 1.0, {a.ip_tokens} image tokens, 1 PNS candidate seed per GPU per step",
                       "parallelism": f"candidates sharded x{world} (no per-step collective)", "outputs_finite": finite,
                       "distributed": {"world_size_reported_by_backend": dist.get_world_size() if world > 1 else 1,
                                       "backend": dist.get_backend() if world > 1 else None,
                                       "per_rank_ms_per_step": per_rank,
                                       "collectives": collectives,
                                       "devices_visible": torch.cuda.device_count()},
                       "conditioning_prepare_ms": {"first_call": cond_ms[0], "later_calls": cond_ms[1:],
                                                   "note": "DenoiseEngine.set_conditioning: K / V caches of the 70 cross-attention layers (140 small GEMMs "
                                                           "+ 20 for the image-prompt tokens), add_embedding; once per image / per PNS run, shared by every "
                                                           "candidate seed; not in `value`"},
                       "value_counts": "denoised latents per second (the reference's output_type='latent'); the VAE decode + "
                                       "post-processing tail (custom_pipelines.py:365-386) is NOT in `value` -- its time is "
                                       "reported under vae_decode",
                       "ms_per_unet_forward": dt / a.steps / a.denoise_steps * 1e3,
                       "clocks_under_load": clocks,
                       "xcd_cells": {"chosen": eng.xcd_cells, "ms_per_step_when_picked": getattr(eng, "xcd_times_ms", None),
                                     "note": "XCD cell shape of the GEMM / conv tile grids, measured once when the plan is recorded "
                                             "(DenoiseEngine._pick_xcd_cells, outside the timed region): 0 = byte-count model, "
                                             "3 = 4 x 2, 2 = 8 x 1 cells over M x N; bit-identical results, box-dependent speed"},
                       "tflop_per_unet_forward": tot_fl / 1e12},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": achieved / 2500.0,
                         "frac_at_sampled_clock": (achieved / (2500.0 * clocks["sclk_mhz"] / 2400.0)) if clocks else None,
                         "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": g_by / n_g,
                         "kernel": "imh::gemm_* (Linear + implicit-GEMM conv3x3 family: gemm_kernel, gemm_dual, gemm_ring, gemm_kg2, gemm_pq, gemm_ws, gemm_w16, conv_halo, conv_hws)",
                         "launches_per_step": n_g, "avg_launch_us": g_ms / n_g * 1e3,
                         "algorithmic_tflop_per_step": g_fl / 1e12,
                         "whole_forward_tflops": tot_fl / (dt / a.steps / a.denoise_steps) / 1e12},
            # second kernel family, same method (HIP events on the launch stream): the flash / decoupled-IP attention
            "roofline_attention": {"bound": "mfma", "achieved": a_fl / (a_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                   "frac": a_fl / (a_ms * 1e-3) / 1e12 / 2500.0, "kernel": "imh::attn_kernel (self-attention)",
                                   "launches_per_step": n_a, "avg_launch_us": a_ms / n_a * 1e3,
                                   "algorithmic_tflop_per_step": a_fl / 1e12,
                                   "note": "head_dim 64: per 64-key tile a wave issues 16 MFMAs (512 cycles of the matrix pipe) and ~120 VALU "
                                           "instructions incl. 32 v_exp_f32 (~600 cycles of the VALU pipe, profiles/r03_valu_mfma_rate_microbench.csv)"},
            # what actually bounds the M = 2048 / 8192 layers: operand bytes through the L2 -> LDS path (LDS-DMA), see lds_operand_bytes
            "roofline_l2_lds": {"bound": "l2->lds", "achieved": g_lds / (g_lds_ms * 1e-3) / 1e12, "unit": "TB/s",
                                "peak": 36.5, "frac": g_lds / (g_lds_ms * 1e-3) / 1e12 / 36.5,
                                "bytes_per_step": g_lds, "kernel": "same GEMM / conv family (dual launches excluded)",
                                "note": "peak = the MEASURED LDS-DMA rate of L2-resident lines, every CU streaming, two workgroups per CU (tools/micro/glds_rate.hip l2: "
                                        "36.5 TB/s = 143 GB/s per CU; 27.8 TB/s with one; profiles/r03_l2_lds_rate_microbench.csv), not a datasheet figure; bytes = "
                                        "operand tile bytes each workgroup stages (lds_operand_bytes).  Inside their K loops the one-workgroup-per-CU launches of "
                                        "the forward run AT this stream's rate for their bytes in flight -- 39 B/clk/CU = 90 GB/s with the 64 x 160 kernel's 84 KB "
                                        "(tools/micro/lds_port.hip, profiles/r05_lds_port_microbench.csv); the rest of a launch is ramp (DESIGN.md 3)"},
        }
        if ip_us:
            kv_fl = 2.0 * 2 * 2 * (77 + a.ip_tokens) * 2048 * 1280          # text + ip K,V projections of one IP-active layer (CFG batch 2)
            per_call = sum(ip_fl) / len(ip_fl) + kv_fl / a.denoise_steps       # K/V counted once per image
            us = sum(ip_us) / len(ip_us)
            res["roofline_ip_attn"] = {
                "bound": "mfma", "achieved": per_call / (us * 1e-6) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                "frac": per_call / (us * 1e-6) / 1e12 / 2500.0,
                "kernel": "imh::xattn_kernel<.., 2, ..> (to_q + norm2 + text SDPA + image-prompt SDPA + axpy, one launch) + "
                          "imh::gemm_kernel (to_out + residual)",
                "calls_per_step": len(ip_us), "avg_call_us": us,
                "algorithmic_gflop_per_call": per_call / 1e9,
                "algorithmic_gflop_per_call_survey_8d": (sum(ip_fl) / len(ip_fl) + kv_fl) / 1e9,
                "note": "one IPAttnProcessor2_0 call on an IP-active layer (attention_processor.py:396-453); the K/V "
                        "projections of the text / image tokens (1.70 GFLOP of SURVEY 8d's 15.97) are step-invariant, "
                        "computed once per image and counted once per image (1/30 per step)"}
            res["roofline_cross_attention"] = {
                "bound": "mfma", "achieved": x_fl / (x_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                "frac": x_fl / (x_ms * 1e-3) / 1e12 / 2500.0, "kernel": "imh::xattn_kernel (all 70 cross-attention layers: to_q + SDPA fused)",
                "launches_per_step": n_x, "avg_launch_us": x_ms / n_x * 1e3, "algorithmic_tflop_per_step": x_fl / 1e12}
        co = [(t[3], m) for t, m in zip(rec.tags, ms) if t[2] == "cross.to_out"]
        if co:
            c_fl, c_ms = sum(f for f, _ in co), sum(m for _, m in co)
            res["roofline_cross_to_out"] = {
                "bound": "mfma", "achieved": c_fl / (c_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": c_fl / (c_ms * 1e-3) / 1e12 / 2500.0,
                "kernel": "imh::gemm_ws_kernel 64 x 160 (to_out + bias + residual of the 70 cross-attention layers: the second launch of an "
                          "IPAttnProcessor2_0 call, attention_processor.py:453)",
                "launches_per_step": len(co), "avg_launch_us": c_ms / len(co) * 1e3,
                "note": "2048 x 1280 x 1280 (60 layers) / 8192 x 640 x 640 (10): one round of 256 workgroups whose K loop streams 573 KB per CU "
                        "through the L2 -> LDS path (tools/micro/lds_port.hip: the loop runs at that stream's rate, 39 B / clk / CU)"}
        if world == 1 and a.stacked > 1:
            try:
                res["roofline_ip_attn_cfg4"] = ip_attn_cfg4(device, dtype)
            except Exception as e:      # noqa: BLE001
                res["roofline_ip_attn_cfg4"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and a.in_flight > 1:
            # extra, NOT the headline: several batch-1 candidates in flight on the one GPU (what PNS does with N > n_gpus)
            try:
                res["concurrent_candidates"] = {"in_flight": a.in_flight, "images_per_sec": run_concurrent(a.in_flight, 2, 1),
                                                "note": "independent batch-1 denoises on separate HIP streams sharing weights and "
                                                        "conditioning; same arithmetic per image as `value`"}
            except Exception as e:      # noqa: BLE001  -- extras must never cost the headline number
                res["concurrent_candidates"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and a.stacked > 1:
            # extra, NOT the headline: the rank's candidates as one UNet batch (PNS with N > n_gpus, configs[4])
            try:
                ips, ok = run_stacked(a.stacked, 2, 1)
                res["stacked_candidates"] = {"per_forward": a.stacked, "images_per_sec": ips, "outputs_finite": ok,
                                             "note": f"UNet batch {2 * a.stacked} (CFG); same arithmetic per image as `value`"}
            except Exception as e:      # noqa: BLE001
                res["stacked_candidates"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and a.stacked > 1 and a.res == 1024:
            # extra, NOT the headline: BASELINE.json configs[3] -- 50 steps, batch 4 per GPU, 16 image tokens from the Resampler, fp16
            try:
                res["configs3_fp16_50steps_batch4_T16"] = configs3_extra(device, a.res)
            except Exception as e:      # noqa: BLE001
                res["configs3_fp16_50steps_batch4_T16"] = {"error": f"{type(e).__name__}: {e}"}
            # ... and configs[4]'s single-GPU shape: 4 candidates per GPU, two 16-token image embeds (T = 32), 30 steps, bf16 attention
            # (fp8 attention declined on measured grounds, profiles/r03_fp8_attention_decision.md)
            try:
                res["configs4_bf16_30steps_batch4_T32"] = configs3_extra(
                    device, a.res, S=4, T=32, steps=30, dtype=torch.bfloat16, n_embeds=2,
                    note="BASELINE.json configs[4] on one GPU (never `value`): dual-category edit, two Resampler image embeds of 16 tokens, "
                         "4 PNS candidates stacked per forward, bf16 attention; parity of this shape: tests/test_gpu_parity_fullsize.py "
                         "(UNet batch 8, T = 32, bf16 vs the CPU oracle)")
            except Exception as e:      # noqa: BLE001
                res["configs4_bf16_30steps_batch4_T32"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and a.stacked > 1:
            try:
                res["box_calibration"] = box_calibration(device, dtype)
            except Exception as e:      # noqa: BLE001
                res["box_calibration"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                res["pns_two_stage"] = pns_two_stage(eng, pipe, unet, (pe, ne, po, no), device, dtype, lat_shape, a.res, stack=a.stacked)
            except Exception as e:      # noqa: BLE001
                res["pns_two_stage"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and a.stacked > 1:        # same "extras" switch: the step right after the path, never in `value`
            try:
                from imagharmony_amd.vae import AutoencoderKL, decode_latents
                zl = out[:1].float() * 0.13025
                per_img = dt / images
                res["vae_decode"], res["value_with_decode"] = {}, {}
                # (a) the reference's precision: a float16 VAE is upcast and decoded in fp32 (custom_pipelines.py:366-372) -> csrc/f32.hip;
                # (b) opt-in: a bfloat16 module decodes in bf16 on the UNet's own kernels (the reference would not upcast it either)
                for label, mdt, prec in (("fp32_reference_precision", torch.float16, "fp32"), ("bf16_native", dtype, "native")):
                    vae = AutoencoderKL().init_random_(1).to(device, mdt)
                    assert vae.precision_for() == prec
                    ms_v = {}
                    for tiled in (False, True):
                        vae.enable_tiling(tiled)
                        img = decode_latents(vae, zl)
                        torch.cuda.synchronize(device)
                        t1 = time.perf_counter()
                        for _ in range(3):
                            img = decode_latents(vae, zl)
                        torch.cuda.synchronize(device)
                        ms_v["tiled" if tiled else "untiled"] = (time.perf_counter() - t1) / 3 * 1e3
                    res["vae_decode"][label] = {"ms_per_image": ms_v, "dtype": "f32" if prec == "fp32" else a.dtype, "module_dtype": str(mdt)[6:],
                                                "outputs_finite": bool(torch.isfinite(img).all().item())}
                    res["value_with_decode"][label] = {"untiled": 1.0 / (per_img + ms_v["untiled"] * 1e-3), "tiled": 1.0 / (per_img + ms_v["tiled"] * 1e-3),
                                                       "unit": "images/sec", "decode_dtype": "f32" if prec == "fp32" else a.dtype}
                    del vae
                res["vae_decode"]["note"] = ("SDXL VAE decoder 128x128 latent -> 1024x1024 image on the HIP kernels (random weights), eager launches; not part of "
                                             "`value`.  fp32_reference_precision = what the reference runs (it upcasts the float16 VAE before vae.decode): fp32 "
                                             "activations / weights / arithmetic (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 matrix rate); bf16_native = the opt-in 16-bit decode")
                res["value_with_decode"]["note"] = ("decoded images per second = `value`'s denoise time + the measured VAE decode of the same latent, serially on this GPU "
                                                    "(the reference returns PIL images: custom_pipelines.py:365-386, test.py:73 enables VAE tiling); quote the "
                                                    "fp32_reference_precision entry as 'decoded like the reference'")
            except Exception as e:      # noqa: BLE001
                res["vae_decode"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and a.stacked > 1:        # same "extras" switch: SURVEY.md 8(a5), once per image, never in `value`
            try:
                res["resampler"] = resampler_extra(device, dtype)
            except Exception as e:      # noqa: BLE001
                res["resampler"] = {"error": f"{type(e).__name__}: {e}"}
        if world > 1:
            # the CPU baseline is a property of the box, timed on rank 0 of the N = 1 run only (it would steal host cores from N ranks'
            # launch threads here); the field stays present so that every line carries it
            res["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": None, "kind": "port",
                                   "sample": "see the N = 1 line of the same round (python bench.py --gpus 1): the fp32 CPU oracle, "
                                             "1 warm-up + 2 timed 1024^2 CFG-2 UNet forwards x 30 steps"}
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(a.cpu_res, a.ip_tokens, a.denoise_steps)
            except Exception as e:      # noqa: BLE001  -- the baseline is informational; never lose the GPU number
                res["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
