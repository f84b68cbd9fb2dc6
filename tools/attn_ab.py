"""GPU tool: interleaved A/B of the attention kernels' variants on the SDXL shapes (one process, rounds interleaved, median and
min per variant; guide rule 24).
  self-attention  imh_debug_set(4, m): 1 in-order key loop, 3 software-pipelined key loop, 5 key-split workgroups
  fused cross     imh_debug_set(3, m): 1 one head per workgroup; 2 / 3 / 4 two heads, 0 / 2 / 4 producer waves, resident key tiles;
                  6 / 7 / 8 the same without the resident key tiles
Usage: python tools/attn_ab.py [--rounds 7] > gpurun_out/attn_ab.json"""
import argparse, json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from imagharmony_amd import lib as L
from imagharmony_amd.ctx import Ctx
from conftest import ref_row_stats

DEV = "cuda:0"


def make_vt(v, n_pad):
    B, n, C_ = v.shape
    vt = torch.zeros(C_, B, n_pad, dtype=v.dtype, device=v.device)
    vt[:, :, :n] = v.permute(2, 0, 1)
    vt = vt.reshape(C_, B * n_pad // 16, 4, 4)
    return vt[:, :, [0, 2, 1, 3], :].reshape(C_, B * n_pad).contiguous()


def make_k(k, n_pad):
    B, n, C_ = k.shape
    kp = torch.zeros(B, n_pad, C_, dtype=k.dtype, device=k.device)
    kp[:, :n] = k
    return kp.view(B, n_pad, C_ // 16, 4, 4)[:, :, :, [0, 2, 1, 3], :].reshape(B, n_pad, C_).contiguous()


def plan_of(emit, dtype, reps):
    """`reps` copies of one launch recorded into a C++ plan: replayed without Python between the launches (a ctypes call per
    launch costs more host time than these kernels run)"""
    rec = Ctx(DEV, dtype, record=True)
    for _ in range(reps):
        emit(rec)
    return rec


def gpu_time(rec, reps):
    """mean GPU time per launch of the recorded plan (HIP events around one replay)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rec.run(); torch.cuda.synchronize()
    e0.record()
    rec.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = L.load()
    dtype = torch.bfloat16
    ctx = Ctx(DEV, dtype)
    out = {}
    # ---- self-attention
    for (B, H, Lq) in [(2, 20, 1024), (2, 10, 4096), (8, 20, 1024), (8, 10, 4096)]:
        C_ = H * 64
        qk = torch.randn(B * Lq, 2 * C_, device=DEV).to(dtype)
        v = torch.randn(B, Lq, C_, device=DEV).to(dtype)
        vt = make_vt(v, Lq)
        o = torch.empty(B * Lq, C_, device=DEV, dtype=dtype)
        rec = plan_of(lambda c: c.attention(qk[:, :C_], qk[:, C_:], vt, o, B, H, Lq, Lq, Lq, 2 * C_, 2 * C_, B * Lq, C_, 0.125), dtype, a.reps)
        res, outs = {1: [], 3: [], 5: []}, {}
        for r in range(a.rounds):
            for m in (1, 3, 5):
                lib.imh_debug_set(4, m)
                res[m].append(gpu_time(rec, a.reps))
                if r == 0:
                    outs[m] = o.float().clone()
        lib.imh_debug_set(4, 0)
        fl = 4.0 * B * H * Lq * Lq * 64
        diff = max(float((outs[1] - outs[3]).abs().max()), float((outs[1] - outs[5]).abs().max()))
        key = f"self B={B} H={H} L={Lq}"
        out[key] = {f"mode{m}": dict(us_median=statistics.median(t), us_min=min(t), tflops=fl / statistics.median(t) / 1e6) for m, t in res.items()}
        out[key]["max_abs_diff_between_modes"] = diff
        print(key, {k: (round(v["us_median"], 1), round(v["tflops"])) for k, v in out[key].items() if k.startswith("mode")}, "diff", diff, file=sys.stderr, flush=True)
    # ---- fused cross-attention (to_q + norm2 + text [+ image-prompt] attention), statistics handed over
    from imagharmony_amd.attention_processor import fold_ln
    for (B, H, Lq, T) in [(2, 20, 1024, 4), (2, 20, 1024, 0), (2, 10, 4096, 0), (8, 20, 1024, 16)]:
        C_ = H * 64
        x = (torch.randn(B * Lq, C_, device=DEV) * 1.3 + 0.5).to(dtype)
        wq = torch.randn(C_, C_) * C_ ** -0.5
        norm = torch.nn.LayerNorm(C_)
        wg, s_, c_ = fold_ln(wq, norm, ctx)
        st = (ref_row_stats(x.float(), C_ // 80).to(DEV), C_ // 80)
        pad = lambda n: (n + 63) // 64 * 64
        k, v = torch.randn(B, 77, C_, device=DEV).to(dtype), torch.randn(B, 77, C_, device=DEV).to(dtype)
        kw = {}
        if T:
            k2, v2 = torch.randn(B, T, C_, device=DEV).to(dtype), torch.randn(B, T, C_, device=DEV).to(dtype)
            kw = dict(k2=make_k(k2, pad(T)), vt2=make_vt(v2, pad(T)), Lk2=T, Lk2_pad=pad(T), ldk2=C_, ldvt2=B * pad(T), scale2=1.0)
        kk, vv = make_k(k, 128), make_vt(v, 128)
        o = torch.empty(B * Lq, C_, device=DEV, dtype=dtype)
        rec = plan_of(lambda c: c.cross_attention(x, wg, kk, vv, o, B, H, Lq, 77, 128, C_, B * 128, 0.125, ln=(s_, c_, 1e-5, st), **kw), dtype, a.reps)
        modes = (1, 3, 4)
        res, outs = {m: [] for m in modes}, {}
        for r in range(a.rounds):
            for m in modes:
                lib.imh_debug_set(3, m)
                res[m].append(gpu_time(rec, a.reps))
                if r == 0:
                    outs[m] = o.float().clone()
        lib.imh_debug_set(3, 0)
        fl = 2.0 * B * Lq * C_ * C_ + 4.0 * B * H * Lq * (77 + T) * 64
        key = f"xattn B={B} H={H} L={Lq} T={T}"
        out[key] = {f"mode{m}": dict(us_median=statistics.median(t), us_min=min(t), tflops=fl / statistics.median(t) / 1e6,
                                     max_abs_diff_vs_mode1=float((outs[m] - outs[1]).abs().max())) for m, t in res.items()}
        print(key, {k: round(v["us_median"], 1) for k, v in out[key].items()}, file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
