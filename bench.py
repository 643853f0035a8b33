#!/usr/bin/env python
"""bench.py -- rays/s of the fused volumetric-render hot path on MI355X (BASELINE.json configs[1]:
ssdnerf_cars_uncond, 128x128 novel-view render of cached triplanes, 251 views/scene, 8 scenes/GPU/batch).

One "step" = BaseNeRF.render of one batch: S scenes x V views x 128x128 rays through
AABB -> bitfield-guided march -> triplane gather -> tiny MLP -> composite -> background blend -> uint8 quantise
(and, for N > 1 GPUs, the RCCL all-gather of the rendered uint8 views).  Inputs (packed triplanes, bitfields, MLP
weights, ray arrays) are resident in HBM before the timed region.  Scenes shard over ranks (weak scaling: S scenes
per rank); there is no collective inside the render.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     : the fused render kernel's ALGORITHMIC bytes (44 B/ray + 288 B/sample) / its mean
                 launch time measured with HIP events on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline : the CPU oracle (reference-shaped loop over the C restatement + PyTorch-CPU decode) timed on the
                 host cores on a bounded sample of the same workload (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
BYTES_PER_RAY = 44             # 24 B in (o, d) + 20 B out (rgb, depth, weights_sum)
BYTES_PER_SAMPLE = 288         # 3 planes x 4 corners x 6 channels x 4 B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scenes", type=int, default=8, help="scenes per GPU per batch (samples_per_gpu in the cars config)")
    ap.add_argument("--views", type=int, default=251, help="views per scene (cars test set: 251)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--variant", default="object", choices=["object", "uniform"])
    ap.add_argument("--plane-dtype", default="float32", choices=["float32", "float16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=64, help="views of scene 0 rendered by the CPU oracle")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd import nerf
    from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
    from ssdnerf_amd.density import get_density

    ns, nv, hw = args.scenes, args.views, args.size
    t_start = time.perf_counter()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)
    params = S.make_decoder_params(2021)
    dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256,
                          plane_dtype=args.plane_dtype)
    dec.load_state_dict(params, strict=False)
    dec = dec.to(dev).eval()

    # ---- resident inputs -----------------------------------------------------------------------------------
    seeds = [2021 + rank * ns + s for s in range(ns)]               # mirrors --diff_seed: distinct scenes per rank
    code_cpu = torch.stack([S.make_triplane(sd, args.variant) for sd in seeds], dim=0)
    code = code_cpu.to(dev)
    log('synthetic scenes built')
    g = torch.Generator().manual_seed(7)
    jit_cpu = [torch.rand(64 ** 3, 3, generator=g) for _ in range(8)]
    grid, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=[j.to(dev) for j in jit_cpu])
    planes = pack_triplanes(code, dec.plane_dtype)
    torch.cuda.synchronize(); log('density grids + packed planes ready')
    poses = S.spiral_poses(nv).to(dev)[None].expand(ns, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(ns, nv, -1).contiguous()
    t0 = time.perf_counter()
    rays_o, rays_d = nerf.get_cam_rays(poses, intr, hw, hw)
    rays_o = rays_o.reshape(ns, nv * hw * hw, 3).contiguous()
    rays_d = rays_d.reshape(ns, nv * hw * hw, 3).contiguous()
    torch.cuda.synchronize()
    ms_raygen = (time.perf_counter() - t0) * 1e3
    log(f'rays generated ({ms_raygen:.0f} ms)')
    n_rays = ns * nv * hw * hw
    # N > 1: every rank ends up with every rank's quantised views (RCCL all-gather over xGMI).  The collective of step i runs on RCCL's
    # stream while step i+1 renders (two landing buffers); the compute stream only waits for it before issuing the next collective.
    gathered = [torch.empty(world * ns, nv, hw, hw, 3, dtype=torch.uint8, device=dev) for _ in range(2)] if world > 1 else None
    pending = {"work": None, "i": 0, "keep": None}
    kernel_events = []

    def step(record=False):
        dec.stage_events = [] if record else None      # HIP events on the launch stream: [before A, between A and B, after B]
        out = dec.render_packed(planes, rays_o, rays_d, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, check_overflow=False)
        if record:
            kernel_events.append(dec.stage_events)
            dec.stage_events = None
        img_u8 = nerf.quantize_u8(out["image"]).reshape(ns, nv, hw, hw, 3)
        if world > 1:
            if pending["work"] is not None:
                pending["work"].wait()
            pending["keep"] = img_u8                     # the source must stay alive until the collective has run
            pending["work"] = dist.all_gather_into_tensor(gathered[pending["i"] & 1], img_u8, async_op=True)
            pending["i"] += 1
        return out, img_u8

    def drain():
        if pending["work"] is not None:
            pending["work"].wait()
            pending["work"] = None

    # one untimed pass for the integer statistics the roofline needs (exact sample count of this workload)
    out = dec.render_packed(planes, rays_o, rays_d, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, want_counts=True, check_overflow=False)
    counts = dec.last_render_stats["sample_counts"]
    n_samples = int(counts.sum().item())
    counts_gt0 = int((counts > 0).sum().item())
    overflow = int(dec.last_render_stats["overflow"].item())
    del counts
    log(f'stat pass done: {n_samples} samples, overflow {overflow}')

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, img_u8 = step(record=True)
    drain()                                              # the last step's collective is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_samples], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        n_samples_all = int(tot.item())
    else:
        n_samples_all = n_samples
    ms_per_step = elapsed / args.steps * 1e3
    log(f'timed region done: {ms_per_step:.2f} ms/step')
    rays_per_s = world * n_rays / (elapsed / args.steps)

    first_hit_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in kernel_events]))
    shade_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in kernel_events]))
    kern_ms = shade_ms
    # dominant kernel = k_shade_queue (gather + MLP + composite).  Its algorithmic bytes: 288 B per sample it shades plus,
    # per hitting ray, 8 B queue entry + 24 B ray + 20 B outputs.
    n_hit = int((counts_gt0))
    algo_bytes = n_samples * BYTES_PER_SAMPLE + n_hit * (8 + BYTES_PER_RAY)
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    first_hit_bytes = n_rays * 24 + (n_rays - n_hit) * 20 + n_hit * 8

    traffic = None          # HBM bytes per launch from the PMC counters: measured offline (separate rocprofv3 passes), valid for the default workload only
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        w = tj["workload"]
        if (w["scenes"], w["views"], w["size"], w["variant"], w["plane_dtype"]) == (ns, nv, hw, args.variant, args.plane_dtype) and tj["kernel"] == "k_shade_mfma":
            traffic = tj["hbm_bytes_per_launch"]
    except Exception:
        pass
    result = {
        "metric": "rays/s (rendered-views/s = rays/s / 16384), SRN Cars 128x128 novel-view render of cached triplanes",
        "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.plane_dtype == "float32" else "f32 math / f16 planes", "data": "synthetic",
        "config": {"workload": "ssdnerf_cars_uncond render of cached triplanes (BASELINE.json configs[1])", "scenes_per_gpu": ns,
                   "views_per_scene": nv, "image": f"{hw}x{hw}", "rays_per_step_per_gpu": n_rays, "grid_size": 64, "max_steps": 256,
                   "T_thresh": 1e-4, "dt_gamma": 0.0, "scene_variant": args.variant, "parallelism": f"scene-parallel x{world}",
                   "collective": "all_gather(uint8 views), overlapped with the next step's render" if world > 1 else "none"},
        "views_per_s": rays_per_s / (hw * hw), "samples_per_s": world * n_samples / (elapsed / args.steps) if world == 1 else n_samples_all / (elapsed / args.steps),
        "mean_samples_per_ray": n_samples / n_rays, "rays_at_step_cap": overflow, "ms_raygen_untimed": ms_raygen,
        "roofline": {"bound": "hbm", "kernel": "k_shade_mfma", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes,
                     "launch_ms": kern_ms, "launches_per_step": 1,
                     "note": "algorithmic = 288 B/sample + 52 B per hitting ray; planes (1.5 MiB/scene) are L2-resident, so real HBM "
                             "traffic is far below this (PMC numbers in DESIGN.md / profiles/)",
                     "other_kernels": {"k_first_hit": {"launch_ms": first_hit_ms, "algorithmic_bytes_per_launch": first_hit_bytes,
                                                       "achieved_GBs": first_hit_bytes / (first_hit_ms * 1e-3) / 1e9}}},
        "hit_rays_per_step_per_gpu": n_hit,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(params, code_cpu[0], bits[0].cpu().numpy(), min(args.cpu_views, nv), hw, out, nv)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(params, code0, bits0, n_views, hw, gpu_out, nv):
    """Oracle on the host cores: scene 0, `n_views` views, same triplane / bitfield / rays as the GPU run.
    Also reports the GPU-vs-oracle error on that sample."""
    import oracle
    from oracle import render as R
    from ssdnerf_amd import synthetic as S
    oracle.build()
    cores = min(os.cpu_count() or 1, 32)   # threads actually used (OpenMP team of the C oracle and torch intra-op pool)
    torch.set_num_threads(cores)
    oracle.set_threads(cores)
    ro, rd = R.get_cam_rays(S.spiral_poses(nv)[:n_views], S.cars_intrinsics(hw, hw)[None].expand(n_views, -1), hw, hw)
    ro, rd = ro.reshape(n_views, -1, 3).numpy(), rd.reshape(n_views, -1, 3).numpy()
    t0 = time.perf_counter()
    imgs = []
    for v in range(n_views):          # the reference renders all views of a scene as one ray batch; per view keeps RAM bounded
        rgb, _, _ = R.render_eval(params, code0, bits0, ro[v], rd[v])
        imgs.append(rgb)
    dt = time.perf_counter() - t0
    ref = np.stack(imgs, 0)
    got = gpu_out["image"][0].reshape(nv, hw * hw, 3)[:n_views].cpu().numpy()
    err = float(np.abs(got - ref).max())
    return {"value": n_views * hw * hw / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"scene 0, {n_views} views of {hw}x{hw} ({n_views * hw * hw} rays), {dt:.1f} s; oracle = C restatement of the reference kernels "
                      f"(OpenMP) + PyTorch-CPU grid_sample/Linear decode, reference-shaped loop",
            "max_abs_rgb_err_gpu_vs_oracle": err}


if __name__ == "__main__":
    main()
