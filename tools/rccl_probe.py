"""GPU tool: execute the RCCL ("nccl") branch of the PNS control path on whatever GPUs the box has, and keep the evidence.
  (1) world size 1 on cuda:0: init_process_group("nccl"), a broadcast / all_gather / all_reduce round trip on device tensors --
      the backend loads, creates its communicator and runs its kernels on the MI355X (a 1-rank collective is a device copy);
  (2) world size 2 with both ranks on the ONE GPU of a gpurun box: what RCCL answers (expected: it refuses duplicate devices) --
      recorded verbatim, so the absence of a 2-GPU number is documented rather than silent;
  (3) world size = visible GPUs when more than one is present (the driver's 8-GPU node): pns.run_pns-style collectives.
Writes one JSON line.  Usage: python tools/rccl_probe.py > gpurun_out/rccl_probe.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import torch.distributed as dist
    from imagharmony_amd import pns
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("PROBE_DEVICE", rank))
    torch.cuda.set_device(dev)
    out = {"rank": rank, "world": world, "device": dev}
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world)
        m = torch.nn.Linear(64, 64).cuda().to(torch.bfloat16)
        with torch.no_grad():
            m.weight.fill_(float(rank + 1))
        out["broadcast_collectives"] = pns.broadcast_module_(m, src=0)
        t = torch.full((1024,), float(rank + 1), device="cuda")
        dist.broadcast(t, src=0)
        g = [torch.empty(4, device="cuda") for _ in range(world)]
        dist.all_gather(g, torch.full((4,), float(rank), device="cuda"))
        s = torch.ones(8, device="cuda")
        dist.all_reduce(s)
        torch.cuda.synchronize()
        out.update(ok=True, weight_after_broadcast=float(m.weight.float().mean()), broadcast_value=float(t[0]),
                   gathered=[float(x[0]) for x in g], all_reduce=float(s[0]), backend=dist.get_backend())
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001 -- the refusal text IS the result
        out.update(ok=False, error=f"{type(e).__name__}: {str(e)[:600]}")
    print("PROBE " + json.dumps(out), flush=True)


def launch(world, same_device):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world + (7 if same_device else 0)), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), PROBE_DEVICE="0" if same_device else str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    res = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            o = "TIMEOUT (120 s)"
        line = [l for l in o.splitlines() if l.startswith("PROBE ")]
        res.append(json.loads(line[0][6:]) if line else {"ok": False, "error": o[-600:]})
    return res


if __name__ == "__main__":
    if "--worker" in sys.argv:
        worker()
        sys.exit(0)
    import torch
    n = torch.cuda.device_count()
    out = {"visible_gpus": n, "world1_nccl": launch(1, False)}
    out["world2_on_one_gpu_nccl"] = launch(2, True)
    if n > 1:
        out[f"world{n}_nccl"] = launch(n, False)
    print(json.dumps(out))
