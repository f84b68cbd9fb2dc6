"""GPU tool: one image's CFG pair as TWO half-batch forwards in flight on two HIP streams (DenoiseEngine cfg_role 0 / 1, the engines
of the CFG-split PNS tail) against the fused batch-2 forward: ms per denoise step, same process, interleaved rounds.
Usage: python tools/cfg_streams_probe.py [--rounds 5] > gpurun_out/cfg_streams.json"""
import argparse, json, os, statistics, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench                                                            # noqa: E402
from imagharmony_amd import lib as L                                   # noqa: E402
from imagharmony_amd import pns                                        # noqa: E402
from imagharmony_amd.denoise import DenoiseEngine                      # noqa: E402
from imagharmony_amd.schedulers import DDIMScheduler                   # noqa: E402

DEV = torch.device("cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    dtype = torch.bfloat16
    L.load()
    unet = bench.build_unet(DEV, dtype, 4)
    pe, ne, po, no = [t.to(DEV) for t in bench.synthetic_conditioning(4)]
    engs = {}
    for role in (None, 0, 1):
        e = DenoiseEngine(unet, DEV, dtype, use_graph=True)
        e.set_conditioning(pe, ne, po, no, 1024, 1024, guidance_scale=5.0, cfg_role=role)
        e.set_schedule(DDIMScheduler(), a.steps)
        engs[role] = e
    z = pns.seed_latents(1, (1, 4, 128, 128)).to(DEV)
    fused = engs[None]
    ref = fused.denoise(z).clone()
    ea, eb = engs[0], engs[1]
    sa, sb = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)

    def split_denoise():
        cur = torch.cuda.current_stream(DEV)
        for e in (ea, eb):
            if e.plan is None:
                e._record()
            e.st.latents.copy_(z.float() * e.init_noise_sigma)
            e.eager.ew(L.EW_STEP_SET, e.st.step, i=(0, 1, 0, 0, 0, 0), descr="step=0")
        for _ in range(a.steps):
            sa.wait_stream(cur); sb.wait_stream(cur)
            with torch.cuda.stream(sa):
                ea.plan.replay()
            with torch.cuda.stream(sb):
                eb.plan.replay()
            cur.wait_stream(sa); cur.wait_stream(sb)
            ea.np_full[0].copy_(ea.noise_pred); ea.np_full[1].copy_(eb.noise_pred)
            ea.plan_tail.replay()
            eb.st.latents.copy_(ea.st.latents)            # (two ranks would each run the tail; one GPU: copy)
            eb.eager.ew(L.EW_STEP_SET, eb.st.step, i=(0, 0, 0, 0, 0, 0), descr="step++")
        return ea.st.latents

    out = split_denoise().clone()
    torch.cuda.synchronize()
    rel = float(((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    res = {"fused": [], "split_two_streams": []}
    for _ in range(a.rounds):
        for name, fn in (("fused", lambda: fused.denoise(z)), ("split_two_streams", split_denoise)):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
    # half-forward alone (what one rank of the CFG-split PNS tail runs per step)
    ea.plan.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ea.plan.replay()
    torch.cuda.synchronize()
    half = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"ms_per_step": {k: statistics.median(v) for k, v in res.items()}, "all": res, "half_forward_alone_ms": half,
                      "rel_rms_split_vs_fused": rel, "steps": a.steps}))


if __name__ == "__main__":
    main()
