"""Preference-guided noise selection (PNS) sharded over the GPUs of one node.

The reference publishes the scheme only as a figure (README.md:27, assets/1.png): N candidate seeds ->
a short preview denoise each -> a judge scores the previews -> the best noise gets the full denoise.
The only code hook is the list-of-seeds generator (ip_adapter/utils.py:83-93).  Candidates are independent
trajectories (same weights, same conditioning, different initial noise), so the path shards with NO
per-step communication:

    rank r of W denoises seeds {i : i mod W == r};
    RCCL (torch.distributed backend "nccl") carries only: the one-time broadcast of the conditioning
    (and optionally of the weights, to guarantee identical replicas), the all_gather of N fp32 scores and
    the broadcast of the winning latent from its owner.

The judge is unpinned upstream (a VLM); ``scorer`` is pluggable and the default is a deterministic
latent-space statistic so that runs are reproducible.  It is NOT a quality claim.
"""
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def default_scorer(latents: torch.Tensor) -> torch.Tensor:
    """[S, 4, h, w] -> [S] deterministic score (higher is better): negative deviation of the per-channel
    standard deviation from 1 -- a placeholder for the reference's VLM judge."""
    s = latents.float().flatten(2).std(dim=2)
    return -(s - 1.0).abs().mean(dim=1)


def shard(seeds: Sequence[int], rank: int, world: int) -> List[int]:
    return [s for i, s in enumerate(seeds) if i % world == rank]


def seed_latents(seed: int, shape) -> torch.Tensor:
    """backend-independent initial noise: drawn on a CPU generator (SURVEY.md 7 'RNG parity')"""
    return torch.randn(tuple(shape), generator=torch.Generator("cpu").manual_seed(int(seed)), dtype=torch.float32)


def broadcast_module_(module: torch.nn.Module, src: int = 0):
    """one-time weight broadcast so every rank holds the same replica (RCCL over xGMI on GPUs)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for p in module.parameters():
        dist.broadcast(p.data, src=src)
    for b in module.buffers():
        dist.broadcast(b.data, src=src)


def broadcast_tensors_(tensors: Sequence[torch.Tensor], src: int = 0):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src=src)


def run_pns(denoise_fn: Callable[[torch.Tensor], torch.Tensor], seeds: Sequence[int], latent_shape,
            scorer: Callable[[torch.Tensor], torch.Tensor] = default_scorer, device="cpu",
            final_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, batch: int = 1):
    """Each rank runs ``denoise_fn(noise [S,4,h,w]) -> latents [S,4,h,w]`` for its share of ``seeds`` (the preview,
    or the full denoise when ``final_fn`` is None), scores them, and the group agrees on the winner.
    ``batch`` = candidates per denoise call (S <= batch): with N > world seeds a rank stacks its candidates into one
    UNet batch (BASELINE.json configs[4]: 4 per GPU -> UNet batch 8), 1.2-1.3x the images/sec of one at a time on
    MI355X; candidates stay independent, so the scores do not depend on ``batch``.

    Returns dict(best_seed, best_score, scores [N], latents = the winner's latents on every rank;
    with ``final_fn`` the winner's noise is re-denoised by its owner rank and that result is returned)."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    seeds = list(seeds)
    mine = shard(seeds, rank, world)
    per = (len(seeds) + world - 1) // world
    local_scores = torch.full((per,), float("-inf"), dtype=torch.float32, device=device)
    local_lat = {}
    for j0 in range(0, len(mine), max(1, int(batch))):
        group = mine[j0:j0 + max(1, int(batch))]
        lat = denoise_fn(torch.cat([seed_latents(s, latent_shape) for s in group], 0))
        sc = scorer(lat)
        for k, s in enumerate(group):
            local_scores[j0 + k] = sc[k].to(device)
            local_lat[s] = lat[k:k + 1].detach().clone()
    if world > 1:
        gathered = [torch.empty_like(local_scores) for _ in range(world)]
        dist.all_gather(gathered, local_scores)
    else:
        gathered = [local_scores]
    scores = torch.full((len(seeds),), float("-inf"), dtype=torch.float32)
    for r in range(world):
        for j, _ in enumerate(shard(seeds, r, world)):
            scores[j * world + r] = gathered[r][j].item()
    best = int(torch.argmax(scores))            # ties -> lowest index: identical on every rank
    owner = best % world
    best_seed = seeds[best]
    if rank == owner:
        out = local_lat[best_seed]
        if final_fn is not None:
            out = final_fn(seed_latents(best_seed, latent_shape)).detach().clone()
        out = out.to(device=device, dtype=torch.float32).contiguous()
    else:
        out = torch.empty((1,) + tuple(latent_shape[1:]), dtype=torch.float32, device=device)
    if world > 1:
        dist.broadcast(out, src=owner)
    return dict(best_seed=best_seed, best_score=float(scores[best]), scores=scores, latents=out, owner=owner)
