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


# ---------------------------------------------------------------------------------------------- world_size 8: the node the north star names
def _worker_node(rank, world, port, n_scenes, per_gpu, q):
    """One rank of an 8-rank scene-parallel evaluation shaped like lib/apis/test.py:12-73 over lib/datasets/samplers/distributed_sampler.py:27-40:
    the rank walks ITS shard of the scene list in batches of ``per_gpu`` scenes (ragged last batch), "renders" them (scene s -> views filled with
    s mod 251, a PSNR-like scalar 20 + s), all ranks exchange the rendered views of the batch, and the logged scalar is reduced at the end."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssdnerf_amd import parallel
    bounds = parallel.shard_bounds(n_scenes, world)
    mine = list(parallel.shard_scenes(n_scenes, rank, world))
    n_batches = max(-(-int(bounds[r + 1] - bounds[r]) // per_gpu) for r in range(world))      # every rank runs the same number of batches (the sampler pads)
    ok, log, sizes = True, [], []
    for b in range(n_batches):
        ids = mine[b * per_gpu:(b + 1) * per_gpu]
        views = (torch.stack([torch.full((1, 2, 2, 3), s % 251, dtype=torch.uint8) for s in ids]) if ids
                 else torch.zeros(0, 1, 2, 2, 3, dtype=torch.uint8))
        equal = all(min(per_gpu, max(0, int(bounds[r + 1] - bounds[r]) - b * per_gpu)) == len(ids) for r in range(world))
        parts = [parallel.all_gather_views(views).reshape(world, len(ids), 1, 2, 2, 3)[r] for r in range(world)] if equal and ids \
            else parallel.all_gather_ragged_views(views)
        for r, part in enumerate(parts):                                       # what every rank holds after the exchange: rank r's scenes of this batch
            want = list(range(int(bounds[r]), int(bounds[r + 1])))[b * per_gpu:(b + 1) * per_gpu]
            ok = ok and part.shape[0] == len(want) and all(int(part[i].min()) == want[i] % 251 == int(part[i].max()) for i in range(len(want)))
        if ids:
            log.append(sum(20.0 + s for s in ids) / len(ids))
            sizes.append(len(ids))
    red = parallel.weighted_log_vars(dict(psnr=log), sizes)
    q.put((rank, mine[0] if mine else -1, len(mine), ok, red["psnr"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_scenes", [704, 701])
def test_eight_rank_scene_parallel_evaluation(n_scenes):
    """The 8-GPU node of north_star on gloo: 704 test scenes (SRN cars) -> the reference's 88-scene shards, 8 scenes per rank per batch, 11 batches;
    701 scenes -> shards of 88 / 87 / 88 / ... with ragged last batches.  Every rank must end up with every rank's views of every batch, and the
    sample-weighted metric must equal the plain mean over all scenes."""
    world, per_gpu = 8, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_node, args=(r, world, port, n_scenes, per_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from ssdnerf_amd.parallel import shard_bounds
    b = shard_bounds(n_scenes, world)
    assert [r[1] for r in res] == [int(v) for v in b[:-1]] and [r[2] for r in res] == [int(b[i + 1] - b[i]) for i in range(world)]
    if n_scenes == 704:
        assert all(r[2] == 88 for r in res)
    assert all(r[3] for r in res)
    want = 20.0 + (n_scenes - 1) / 2
    assert all(abs(r[4] - want) < 1e-3 for r in res), ([r[4] for r in res], want)


# ---------------------------------------------------------------------------------------------- the second axis: views of one scene over the ranks (#scenes < #GPUs)
def test_render_shard_plan_covers_every_view_once():
    from ssdnerf_amd.parallel import plan_render_shards
    for scenes, views, world in [(1, 251, 8), (3, 251, 8), (8, 251, 8), (11, 10, 8), (2, 250, 8), (1, 5, 8), (7, 251, 2), (5, 3, 8)]:
        plan = plan_render_shards(scenes, views, world)
        assert len(plan) == world
        cover = np.zeros((scenes, views), dtype=int)
        for a, b, c, d in plan:
            cover[a:b, c:d] += 1
        assert (cover == 1).all(), (scenes, views, world, plan)
    # one scene on the node of north_star: 251 views -> 31 / 32 per rank, contiguous
    assert plan_render_shards(1, 251, 8) == [(0, 1, 0, 31), (0, 1, 31, 63), (0, 1, 63, 94), (0, 1, 94, 126), (0, 1, 126, 157), (0, 1, 157, 188), (0, 1, 188, 220), (0, 1, 220, 251)]
    # scenes >= ranks: the reference's scene partition, all views
    assert plan_render_shards(704, 251, 8)[3] == (264, 352, 0, 251)


def _worker_views(rank, world, port, scenes, views, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssdnerf_amd import parallel

    def render(a, b, c, d):                                                    # view v of scene s: every pixel (s * 16 + v) % 251, plus a pixel that names the rank
        out = torch.zeros(b - a, d - c, 2, 2, 3, dtype=torch.uint8)
        for i, s in enumerate(range(a, b)):
            for j, v in enumerate(range(c, d)):
                out[i, j] = (s * 16 + v) % 251
        return out
    full = parallel.render_sharded(render, scenes, views)
    want = torch.tensor([[(s * 16 + v) % 251 for v in range(views)] for s in range(scenes)], dtype=torch.uint8)
    ok = tuple(full.shape) == (scenes, views, 2, 2, 3) and bool((full == want[:, :, None, None, None]).all())
    q.put((rank, ok, parallel.plan_render_shards(scenes, views, world)[rank]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenes,views", [(1, 251), (3, 10), (8, 4)])
def test_eight_rank_view_split_render(scenes, views):
    """One scene's 251 views over the 8 ranks of the node (31 / 32 views each), three scenes over 8 ranks (3 + 2 + 3 ranks), and the scene-parallel case through the
    same entry point: every rank ends up with every view of every scene."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_views, args=(r, world, port, scenes, views, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    if scenes == 1:
        assert [r[2][3] - r[2][2] for r in res] == [31, 32, 31, 32, 31, 31, 32, 31]
