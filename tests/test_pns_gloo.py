"""CPU, world_size 2 over gloo: the PNS sharding / gather / winner-broadcast logic (imagharmony_amd.pns).
The denoiser is a deterministic stand-in (the HIP engine needs a GPU); what is tested is the N>1 path:
seed sharding, score all_gather order, identical winner on every rank, latent broadcast from the owner."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imagharmony_amd import pns


def _fake_denoise(noise):
    return noise * 0.5 + noise.mean(dim=(1, 2, 3), keepdim=True)      # per candidate: batching must not mix them


def _worker(rank, world, port, seeds, q, batch=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lin = torch.nn.Linear(4, 4)
        with torch.no_grad():
            lin.weight.fill_(float(rank + 1))
        pns.broadcast_module_(lin, src=0)
        r = pns.run_pns(_fake_denoise, seeds, (1, 4, 8, 8), final_fn=lambda n: _fake_denoise(n) + 1.0, batch=batch)
        q.put((rank, r["best_seed"], r["scores"].tolist(), r["latents"].sum().item(), lin.weight[0, 0].item(), r["owner"]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest


@pytest.mark.parametrize("batch", [1, 2])
def test_pns_world2_matches_single_process(batch):
    """batch = candidates stacked per denoise call on a rank (configs[4]: 4 per GPU): same scores, same winner"""
    seeds = [11, 7, 3, 19, 5, 23, 2]
    single = pns.run_pns(_fake_denoise, seeds, (1, 4, 8, 8), final_fn=lambda n: _fake_denoise(n) + 1.0)
    batched = pns.run_pns(_fake_denoise, seeds, (1, 4, 8, 8), final_fn=lambda n: _fake_denoise(n) + 1.0, batch=3)
    assert batched["best_seed"] == single["best_seed"] and torch.equal(batched["scores"], single["scores"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, seeds, q, batch)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, best, scores, lat_sum, w00, owner in res:
        assert best == single["best_seed"]
        assert scores == single["scores"].tolist()
        assert abs(lat_sum - single["latents"].sum().item()) < 1e-4
        assert w00 == 1.0                                  # weights broadcast from rank 0
        assert owner == seeds.index(best) % 2


def test_shard_and_seed_noise():
    assert pns.shard([0, 1, 2, 3, 4], 0, 2) == [0, 2, 4] and pns.shard([0, 1, 2, 3, 4], 1, 2) == [1, 3]
    assert pns.shard([], 0, 2) == []
    a, b = pns.seed_latents(5, (1, 4, 8, 8)), pns.seed_latents(5, (1, 4, 8, 8))
    assert torch.equal(a, b) and not torch.equal(a, pns.seed_latents(6, (1, 4, 8, 8)))
    s = pns.default_scorer(torch.randn(3, 4, 8, 8))
    assert s.shape == (3,)
