"""CPU, world_size 2 / 4 / 8 over gloo: the PNS sharding / gather / winner-broadcast logic (imagharmony_amd.pns).
The denoiser is a deterministic stand-in (the HIP engine needs a GPU); what is tested is the N>1 path:
seed sharding (also N < W: idle ranks; N = 32 stacked 4 per call), score all_gather order, identical winner on every rank, latent
broadcast from the owner, the two-stage final denoise on the owner, and the CFG-split final denoise shared by the owner and the
next rank (one all_gather of the two noise-prediction halves per step inside the pair's group)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imagharmony_amd import pns


def _fake_denoise(noise):
    return noise * 0.5 + noise.mean(dim=(1, 2, 3), keepdim=True)      # per candidate: batching must not mix them


def _tiny_judge():
    """the default CLIP-space judge on a tiny random CLIP vision model (transformers) and a stand-in decoder"""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(0)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                                                          num_attention_heads=2, image_size=16, patch_size=8,
                                                          projection_dim=24)).eval()
    decode = lambda z: torch.tanh(torch.nn.functional.interpolate(z[:, :3], scale_factor=4.0, mode="nearest"))
    target = torch.randn(1, 24, generator=torch.Generator().manual_seed(5))
    return pns.ClipPreferenceJudge(decode, clip, target)


def _worker(rank, world, port, seeds, q, batch=1, judge=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
        with torch.no_grad():
            for p in net.parameters():
                p.fill_(float(rank + 1))
        n_coll = pns.broadcast_module_(net, src=0, bucket_bytes=64)     # tiny buckets: several flat collectives
        assert 1 < n_coll < len(list(net.parameters())) + len(list(net.buffers()))
        assert all(bool((p == 1.0).all()) for p in net.parameters())
        scorer = _tiny_judge() if judge else pns.default_scorer
        r = pns.run_pns(_fake_denoise, seeds, (1, 4, 8, 8), scorer=scorer, final_fn=lambda n: _fake_denoise(n) + 1.0, batch=batch)
        q.put((rank, r["best_seed"], r["scores"].tolist(), r["latents"].sum().item(), net[0].weight[0, 0].item(), r["owner"]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest


@pytest.mark.parametrize("batch,judge", [(1, False), (2, False), (2, True)])
def test_pns_world2_matches_single_process(batch, judge):
    """batch = candidates stacked per denoise call on a rank (configs[4]: 4 per GPU): same scores, same winner;
    judge=True: the default CLIP-space judge (decode -> CLIP embedding -> cosine to the fused target) as the scorer"""
    seeds = [11, 7, 3, 19, 5, 23, 2]
    scorer = _tiny_judge() if judge else pns.default_scorer
    single = pns.run_pns(_fake_denoise, seeds, (1, 4, 8, 8), scorer=scorer, final_fn=lambda n: _fake_denoise(n) + 1.0)
    batched = pns.run_pns(_fake_denoise, seeds, (1, 4, 8, 8), scorer=scorer, final_fn=lambda n: _fake_denoise(n) + 1.0, batch=3)
    assert batched["best_seed"] == single["best_seed"]
    assert torch.allclose(batched["scores"], single["scores"], atol=1e-6)
    if judge:
        assert single["scores"].abs().max() <= 1.0 + 1e-5 and single["scores"].unique().numel() == len(seeds)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, seeds, q, batch, judge)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, best, scores, lat_sum, w00, owner in res:
        assert best == single["best_seed"]
        assert torch.allclose(torch.tensor(scores), single["scores"], atol=1e-6)
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


# ---------------------------------------------------------------------------------------------------------------------
# worlds 4 and 8; N < W; N = 32 stacked; CFG-split final denoise
STEPS, GUIDE = 6, 5.0


def _half(lat, role):
    """stand-in for one half of the CFG pair's UNet forward (role 0 unconditional, 1 conditional)"""
    return lat * (0.1 if role == 0 else 0.2) + (0.0 if role == 0 else 0.3) * lat.mean(dim=(1, 2, 3), keepdim=True)


def _fused_final(noise):
    lat = noise.clone()
    for _ in range(STEPS):
        un, co = _half(lat, 0), _half(lat, 1)
        lat = lat - 0.05 * (un + GUIDE * (co - un))
    return lat


def _split_final(noise, exchange, role):
    lat = noise.clone()
    for _ in range(STEPS):
        un, co = exchange(_half(lat, role))
        lat = lat - 0.05 * (un + GUIDE * (co - un))
    return lat


def _worker_n(rank, world, port, seeds, q, batch, split):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = pns.pair_groups() if split else None
        calls = []

        def preview(noise):
            calls.append(noise.shape[0])
            return _fake_denoise(noise)
        r = pns.run_pns(preview, seeds, (1, 4, 8, 8), final_fn=_fused_final, batch=batch,
                        final_split_fn=_split_final if split else None, pairs=pairs)
        # (by value: a torch tensor in a multiprocessing queue travels as a file descriptor the receiver fetches from THIS process -- an
        # EOFError in the parent whenever the worker has exited first, one run in three of the world-8 case on a loaded box)
        q.put((rank, r["best_seed"], r["scores"].tolist(), r["latents"].clone().numpy(), r["owner"], calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_seeds,batch,split", [(4, 3, 1, False), (4, 32, 4, False), (4, 6, 1, True), (8, 5, 1, True), (8, 32, 4, False), (2, 4, 1, True)])
def test_pns_wider_worlds_idle_ranks_stacking_and_cfg_split_final(world, n_seeds, batch, split):
    seeds = [(7 * i + 3) % 101 for i in range(n_seeds)]
    single = pns.run_pns(_fake_denoise, seeds, (1, 4, 8, 8), final_fn=_fused_final)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, seeds, q, batch, split)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    best_idx = seeds.index(single["best_seed"])
    for rank, best, scores, lat, owner, calls in res:
        lat = torch.from_numpy(lat)
        assert best == single["best_seed"] and owner == best_idx % world
        assert torch.allclose(torch.tensor(scores), single["scores"], atol=1e-6)
        # the split final (two ranks, halves exchanged every step) reproduces the fused one: same arithmetic on the same values
        assert torch.allclose(lat, single["latents"], atol=1e-5), (rank, (lat - single["latents"]).abs().max())
        mine = pns.shard(seeds, rank, world)
        assert sum(calls) == len(mine) and (not mine or max(calls) <= batch)         # idle ranks (N < W) denoise nothing
        if n_seeds < world and rank >= n_seeds:
            assert calls == []
