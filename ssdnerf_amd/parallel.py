"""Scene-parallel multi-GPU helpers: one process per GPU, ``torch.distributed`` (backend ``nccl`` == RCCL on ROCm, over
xGMI; ``gloo`` on CPU for tests).  The hot path shards by independent scenes: no collective inside
render or DDIM, one all-gather of the rendered uint8 views per batch.

``shard_scenes`` keeps the reference's partition (lib/datasets/samplers/distributed_sampler.py:27-40 and the code-cache
split lib/models/autodecoders/multiscene_nerf.py:44-48: ``round(linspace(0, n, world + 1))``) so that the same scene lands
on the same rank."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(num_scenes: int, world_size: int) -> np.ndarray:
    return np.round(np.linspace(0, num_scenes, num=world_size + 1)).astype(np.int64)


def shard_scenes(num_scenes: int, rank: int, world_size: int) -> range:
    b = shard_bounds(num_scenes, world_size)
    return range(int(b[rank]), int(b[rank + 1]))


def all_gather_views(views_u8: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """(S_local, V, H, W, 3) uint8 on every rank -> (world * S_local, V, H, W, 3), rank-major.  Equal S_local per rank
    (pad the last batch upstream); with xGMI's all-pairs links RCCL runs this as a direct one-hop gather."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return views_u8
    world = dist.get_world_size(group)
    views_u8 = views_u8.contiguous()
    out = torch.empty((world * views_u8.size(0),) + tuple(views_u8.shape[1:]), dtype=views_u8.dtype, device=views_u8.device)
    dist.all_gather_into_tensor(out, views_u8, group=group)
    return out


def all_gather_ragged_views(views_u8: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> List[torch.Tensor]:
    """Variant for unequal per-rank scene counts (the tail of a scene list): gathers sizes first, pads to the maximum."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [views_u8]
    world = dist.get_world_size(group)
    n = torch.tensor([views_u8.size(0)], dtype=torch.int64, device=views_u8.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(views_u8.shape[1:]), dtype=views_u8.dtype, device=views_u8.device)
    pad[: views_u8.size(0)] = views_u8
    out = torch.empty((world * m,) + tuple(views_u8.shape[1:]), dtype=views_u8.dtype, device=views_u8.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * m: r * m + sizes[r]] for r in range(world)]


def reduce_mean(x: torch.Tensor) -> torch.Tensor:
    """lib/core/utils/misc.py:34-40: all_reduce(SUM) of x / world."""
    if not (dist.is_available() and dist.is_initialized()):
        return x
    x = x.clone().div_(dist.get_world_size())
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    return x


def weighted_log_vars(log_vars: dict, batch_sizes: List[int], device=None) -> dict:
    """The closing reduction of a scene-parallel evaluation (lib/apis/test.py:58-73): every logged scalar is averaged over ALL scenes of all
    ranks, weighted by the number of scenes of the batch it came from -- sum_ranks(sum_batches(value * n)) / sum_ranks(sum_batches(n)) -- with one
    scalar all_reduce per key plus one for the denominator.  ``log_vars``: key -> list of per-batch values of THIS rank; ``batch_sizes``: scenes per
    batch of this rank (ranks may hold different numbers of batches and ragged last batches)."""
    n = torch.tensor(batch_sizes, dtype=torch.float, device=device)
    total = n.sum()
    distributed = dist.is_available() and dist.is_initialized()
    if distributed:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    out = {}
    for key, values in log_vars.items():
        acc = (torch.tensor([float(v) for v in values], dtype=torch.float, device=device) * n).sum()
        if distributed:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        out[key] = float(acc / total)
    return out


# ---------------------------------------------------------------------------------------------- second axis: views ("#scenes < #GPUs")
def plan_render_shards(num_scenes: int, num_views: int, world_size: int) -> List[Tuple[int, int, int, int]]:
    """Per rank ``(scene_lo, scene_hi, view_lo, view_hi)`` of a render of ``num_scenes`` scenes x ``num_views`` views on ``world_size`` ranks.
    ``num_scenes >= world_size``: the reference's scene partition, every rank renders all views of its scenes (``shard_scenes``).
    ``num_scenes < world_size`` (single-scene latency, the tail of a scene list): the RANKS are partitioned over the scenes with the same
    ``round(linspace)`` rule and the ranks of one scene split its views the same way -- rays are independent (the reference chunks them freely,
    lib/models/autodecoders/base_nerf.py:506-512), the 1.2 MB scene is replicated on its ranks.  A rank may get an empty view range only if a scene
    has fewer views than ranks."""
    if num_scenes >= world_size:
        b = shard_bounds(num_scenes, world_size)
        return [(int(b[r]), int(b[r + 1]), 0, num_views) for r in range(world_size)]
    rb = shard_bounds(world_size, num_scenes)                      # ranks [rb[s], rb[s+1]) render scene s
    plan = []
    for s in range(num_scenes):
        k = int(rb[s + 1] - rb[s])
        vb = shard_bounds(num_views, k)
        plan += [(s, s + 1, int(vb[j]), int(vb[j + 1])) for j in range(k)]
    return plan


def all_gather_render_shards(views_u8: torch.Tensor, num_scenes: int, num_views: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Every rank holds the uint8 views of ITS shard of ``plan_render_shards`` -- (scene_hi - scene_lo, view_hi - view_lo, H, W, 3) -- and gets the
    whole render (num_scenes, num_views, H, W, 3).  One ``all_gather_into_tensor`` of shards padded to the largest one (views differ by at most one
    between the ranks of a scene; scenes by at most one between ranks), then a local reassembly."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return views_u8
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    plan = plan_render_shards(num_scenes, num_views, world)
    lo_s, hi_s, lo_v, hi_v = plan[rank]
    assert tuple(views_u8.shape[:2]) == (hi_s - lo_s, hi_v - lo_v), (tuple(views_u8.shape), plan[rank])
    ms = max(p[1] - p[0] for p in plan)
    mv = max(p[3] - p[2] for p in plan)
    tail = tuple(views_u8.shape[2:])
    pad = views_u8.new_zeros((ms, mv) + tail)
    pad[: hi_s - lo_s, : hi_v - lo_v] = views_u8
    out = views_u8.new_empty((world, ms, mv) + tail)
    dist.all_gather_into_tensor(out.view((world * ms, mv) + tail), pad, group=group)
    full = views_u8.new_empty((num_scenes, num_views) + tail)
    for r, (a, b, c, d) in enumerate(plan):
        full[a:b, c:d] = out[r, : b - a, : d - c]
    return full


def render_sharded(render_fn, num_scenes: int, num_views: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """``render_fn(scene_lo, scene_hi, view_lo, view_hi) -> uint8 views`` of that block (e.g. ``nerf.render(..., return_u8=True)[2]`` on the sliced codes, poses and
    intrinsics); returns the whole (num_scenes, num_views, H, W, 3) on every rank.  No collective but the closing all-gather."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    a, b, c, d = plan_render_shards(num_scenes, num_views, world)[rank]
    mine = render_fn(a, b, c, d)
    return all_gather_render_shards(mine, num_scenes, num_views, group)
