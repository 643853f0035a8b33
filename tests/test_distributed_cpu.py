"""world_size-2 gloo coverage of the N>1 path: scene sharding (reference partition), all-gather of rendered views
(equal and ragged shards) and the sample-weighted metric reduction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_scenes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssdnerf_amd import parallel
    mine = parallel.shard_scenes(n_scenes, rank, world)
    # "render": scene s produces views filled with the value s (uint8), 2 views of 4x4
    views = torch.stack([torch.full((2, 4, 4, 3), s, dtype=torch.uint8) for s in mine]) if len(mine) else torch.zeros(0, 2, 4, 4, 3, dtype=torch.uint8)
    parts = parallel.all_gather_ragged_views(views)
    got = torch.cat(parts, dim=0)
    ok_ragged = got.shape[0] == n_scenes and all(int(got[s].min()) == s == int(got[s].max()) for s in range(n_scenes))
    eq = torch.full((3, 2, 4, 4, 3), rank, dtype=torch.uint8)
    g = parallel.all_gather_views(eq)
    ok_equal = g.shape[0] == 3 * world and all(int(g[r * 3:(r + 1) * 3].float().mean()) == r for r in range(world))
    m = parallel.reduce_mean(torch.tensor([float(rank + 1)]))
    ok_mean = abs(float(m) - (world + 1) / 2) < 1e-6
    q.put((rank, list(mine), ok_ragged, ok_equal, ok_mean))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_scenes", [7, 8])
def test_two_rank_scene_parallel(n_scenes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_scenes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    covered = sum((r[1] for r in res), [])
    assert covered == list(range(n_scenes))                       # disjoint, ordered, complete
    assert all(r[2] and r[3] and r[4] for r in res)


def test_shard_matches_reference_partition():
    from ssdnerf_amd.parallel import shard_bounds
    # round(linspace(0, n, ws+1)) as in the reference's sampler / code cache split
    assert shard_bounds(704, 8).tolist() == [0, 88, 176, 264, 352, 440, 528, 616, 704]
    assert shard_bounds(10, 4).tolist() == [0, 2, 5, 8, 10]


def _worker_train_state(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODELS
    from ssdnerf_amd.diffusion import DDPMMSELossMod
    from ssdnerf_amd.models import NormalizedTanhCode
    m = MODELS.build(dict(type="MultiSceneNeRF", code_size=(3, 6, 8, 8), grid_size=16, cache_size=7,
                          decoder=dict(type="TriPlaneDecoder", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64]),
                          train_cfg=dict(optimizer=dict(type="Adam", lr=0.01))))
    # training-time statistics are averaged over ranks: the prior loss' running norm and the code activation's running moments
    loss = DDPMMSELossMod(rescale_mode=None, data_info=dict(pred="v_t_pred", target="v_t"), scale_norm=True, momentum=0.5).train()
    x0 = torch.full((2, 4, 2, 2), float(rank + 1))
    loss(dict(v_t_pred=x0, v_t=x0 * 0, x_0=x0, timesteps=torch.tensor([1, 2])))
    act = NormalizedTanhCode(std=0.5, momentum=0.5).train()
    act(torch.full((2, 3), float(rank)) + torch.tensor([[-1.0, 0.0, 1.0]]), update_stats=True)
    q.put((rank, sorted(m.cache), float(loss.norm_factor), float(act.running_mean), float(act.running_var)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_cache_shards_and_averaged_training_statistics():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_train_state, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3] and res[1][1] == [4, 5, 6]              # round(linspace(0, 7, 3)) = 0, 4, 7: the reference's cache split
    # norm_factor: 0.5 * 1 + 0.5 * mean_ranks(mean(x0^2)) = 0.5 + 0.5 * (1 + 4) / 2, identical on both ranks
    assert res[0][2] == pytest.approx(1.75) and res[1][2] == pytest.approx(1.75)
    # running_mean: 0.5 * mean_ranks(rank) = 0.25; running_var: 0.5 * 0.25 + 0.5 * mean_ranks(var) with var = 0.8 on both ranks
    assert res[0][3] == pytest.approx(0.25) and res[1][3] == pytest.approx(0.25)
    assert res[0][4] == pytest.approx(0.5 * 0.25 + 0.5 * 0.8) and res[1][4] == pytest.approx(res[0][4])
