#!/usr/bin/env python
"""bench.py -- rays/s of the fused volumetric-render hot path on MI355X (BASELINE.json configs[1]:
ssdnerf_cars_uncond, 128x128 novel-view render of cached triplanes, 251 views/scene, 8 scenes/GPU/batch).

One "step" = BaseNeRF.render of one batch: S scenes x V views x 128x128 rays through
camera -> ray -> AABB -> bitfield-guided march -> triplane gather -> tiny MLP -> composite -> background blend -> uint8 quantise (in the same kernels)
(and, for N > 1 GPUs, the RCCL all-gather of the rendered uint8 views).  Inputs (packed triplanes, bitfields, MLP
weights, camera poses + intrinsics) are resident in HBM before the timed region; rays are generated inside the kernels
(`--ray-arrays` feeds pre-materialised (S,N,3) arrays instead, the reference API's form).  Scenes shard over ranks (weak
scaling: S scenes per rank); there is no collective inside the render.

`python bench.py --gpus N` without a torchrun environment spawns its own N ranks (torch.distributed.run, 127.0.0.1).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      : the shading kernel's ALGORITHMIC bytes (288 B/sample + per-hitting-ray bytes) / its mean launch
                  time measured with HIP events on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline  : the CPU oracle (reference-shaped loop over the C restatement + PyTorch-CPU decode) timed on the
                  host cores on a bounded sample of the same workload (rank 0, N == 1 only).
  gpu_baseline  : "B1" of BASELINE.md -- the reference-shaped path on this GPU: the <=256-iteration alive-ray loop of
                  base_volume_renderer.py:79-123 over the UNFUSED operators with the eager PyTorch decode (grid_sample + nn.Linear) and a
                  device->host sync per iteration, on a bounded sample of the same workload; value / gpu_baseline.value is the speed-up.
  ddim          : the DDIM leg of the same config (50 steps over 8 scenes' triplane latents, cars UNet, V-prediction): ms per step,
                  TFLOP/s and the fraction of the 2.5 PFLOP/s dense bf16 MFMA peak, for the config's fp32 executor and the bf16 one.
  sampling      : north_star's scenes/s: noise -> 50-step DDIM -> density grids -> 251-view render (-> all-gather at N > 1), fp32 and bf16; at
                  N > 1 every rank runs it and `scenes_per_s` (also copied to the top level) is the node's aggregate.
  recons        : config 3: ms per rendering-guided DDIM step, ms per fine-tuning iteration, the 75 + 25 schedule projected from them.
  boundary_rays : termination tests that landed within 2e-6 of T_thresh in one step (the only rays whose integer sample count may
                  differ from the reference's), and the exact sample total, so a drift between kernel builds is visible.
  uniform_variant: the same render on the worst-case fog scenes (every ray marches through occupied space).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
MFMA_PEAK_TFLOPS = 2500.0      # same guide: dense bf16 MFMA peak (no sparsity)
BYTES_PER_SAMPLE = 288         # 3 planes x 4 corners x 6 channels x 4 B
UNET_FLOP_PER_SCENE = 2.18e11  # cars UNet forward


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 10 untimed + 20 timed steps (0.2 s).  With 2 + 5 the timed region still sits on the clock ramp of a cold device: 5.94-6.07 ms per step
    # against 5.79-5.87 ms with 10 + 20 and 5.79 ms with 20 + 50 in one session on one box (profiles/r03/z_warmup_ab.txt)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scenes", type=int, default=8, help="scenes per GPU per batch (samples_per_gpu in the cars config)")
    ap.add_argument("--views", type=int, default=251, help="views per scene (cars test set: 251)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--variant", default="object", choices=["object", "uniform"])
    ap.add_argument("--plane-dtype", default="float32", choices=["float32", "float16"])
    ap.add_argument("--ray-arrays", action="store_true", help="feed materialised (S,N,3) ray arrays instead of cameras")
    ap.add_argument("--packed-loop", action="store_true", help="time TriPlaneDecoder.render_packed(check_overflow=False) instead of nerf.render (the r01-r03 "
                    "timed region: no overflow-flag read per batch); A/B only")
    ap.add_argument("--sync-overflow-check", action="store_true", help="read every render's overflow flag before the next render is queued (the r04 timed "
                    "step) instead of nerf.render's deferred check (flag copied asynchronously, examined behind the next render's launches)")
    ap.add_argument("--no-prefetch", action="store_true", help="do not launch the next step's stage A beside this step's shading kernel (nerf.render's next_batch, r06): the r05 step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-nccl", action="store_true", help="with --gpus 1: take the N > 1 code path with a ONE-rank RCCL process group (backend 'nccl', "
                    "world_size 1) -- RCCL initialisation, the asynchronous all-gather on RCCL's stream, graph captures beside RCCL's watchdog thread -- a "
                    "self-check of the multi-GPU path on a one-GPU box, not a measurement of scaling")
    ap.add_argument("--no-extras", action="store_true", help="skip gpu_baseline / ddim / uniform_variant (profiling runs)")
    ap.add_argument("--cpu-views", type=int, default=251, help="views of scene 0 rendered by the CPU oracle (251 = the whole scene, ~10 s on 32 host threads)")
    ap.add_argument("--b1-views", type=int, default=251, help="views of scene 0 rendered by the reference-shaped eager GPU path (the reference batches all views of a scene)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-full-recons", action="store_true", help="skip the two whole reconstruction batches (configs[2] as shipped: ~4 s; configs[4]'s "
                    "Langevin x guidance schedule under bf16 + fp16 planes: ~20 s)")
    return ap.parse_args()


def self_spawn(args):
    """--gpus N outside a torchrun environment: launch N ranks of this script (one process per GPU over RCCL) and relay their output."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # SSDNERF_BENCH_SHARE_DEVICE=1 (test aid, never a measurement): every rank on cuda:0 over gloo, so that the N > 1 control flow can be exercised on
    # a one-GPU box (RCCL refuses two ranks on one device); the JSON line then says so in config.parallelism
    share_device = os.environ.get("SSDNERF_BENCH_SHARE_DEVICE", "0") == "1" and world > 1
    if share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.dry_nccl                      # the N > 1 code path (collectives, per-rank breakdown, every rank runs the legs)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

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
    g = torch.Generator().manual_seed(7)
    jit_cpu = [torch.rand(64 ** 3, 3, generator=g) for _ in range(8)]
    poses = S.spiral_poses(nv).to(dev)[None].expand(ns, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(ns, nv, -1).contiguous()
    n_rays = ns * nv * hw * hw
    bytes_per_hit_ray = 8 + 20 + (24 if args.ray_arrays else 0)     # queue entry + outputs (+ the ray, when it is read instead of generated)

    def build_scenes(variant):
        seeds = [2021 + rank * ns + s for s in range(ns)]           # mirrors --diff_seed: distinct scenes per rank
        code_cpu = torch.stack([S.make_triplane(sd, variant) for sd in seeds], dim=0)
        code = code_cpu.to(dev)
        _, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=[j.to(dev) for j in jit_cpu])
        return code_cpu, pack_triplanes(code, dec.plane_dtype), bits

    code_cpu, planes, bits = build_scenes(args.variant)
    torch.cuda.synchronize(); log("synthetic scenes, density grids, packed planes ready")
    rays = None
    if args.ray_arrays:
        ro, rd = nerf.get_cam_rays(poses, intr, hw, hw)
        rays = (ro.reshape(ns, -1, 3).contiguous(), rd.reshape(ns, -1, 3).contiguous())
        del ro, rd

    def render(planes_, bits_, **kw):
        if rays is not None:
            return dec.render_packed(planes_, rays[0], rays[1], bits_, 64, [0.0] * ns, 1e-4, bg_color=1.0, check_overflow=False, **kw)
        return dec.render_packed(planes_, None, None, bits_, 64, [0.0] * ns, 1e-4, bg_color=1.0, check_overflow=False, cams=(poses, intr, hw, hw), want_u8=True,
                                 **kw)

    def render_product(planes_, bits_, code_):
        """the timed step since r04: ``nerf.render`` = ``BaseNeRF.render`` on the cached (packed) planes -- the same two launches as
        ``render_packed`` plus what the product path does around them: the ONE overflow-flag read per batch (a host sync) that decides whether the batch
        has to be redone through the stepwise path, and the uint8 views"""
        image, depth, image_u8 = nerf.render(dec, code_, bits_, hw, hw, intr, poses, grid_size=64, bg_color=1.0, cfg={}, planes=planes_,
                                              rays=rays, return_u8=True, defer_overflow_check=not args.sync_overflow_check,
                                              next_batch=None if args.no_prefetch else (bits_, intr, poses))     # (r06) the next step's stage A beside this step's shading kernel
        return {"image": image.reshape(ns, nv * hw * hw, 3), "depth": depth.reshape(ns, nv * hw * hw), "image_u8": image_u8}

    # N > 1: every rank ends up with every rank's quantised views (RCCL all-gather over xGMI).  The collective of step i runs on RCCL's
    # stream while step i+1 renders (two landing buffers); the compute stream only waits for it before issuing the next collective.
    gathered = [torch.empty(world * ns, nv, hw, hw, 3, dtype=torch.uint8, device=dev) for _ in range(2)] if multi else None
    pending = {"work": None, "i": 0, "keep": None, "waits": []}   # waits: HIP event pairs around every wait of the compute stream for a collective

    def step(planes_, bits_, events=None, code_=None):
        dec.stage_events = [] if events is not None else None      # HIP events on the launch stream: [before A, between A and B, after B]
        out = render(planes_, bits_) if args.packed_loop or code_ is None else render_product(planes_, bits_, code_)
        if events is not None:
            events.append(dec.stage_events)
            dec.stage_events = None
        # the uint8 views that are gathered / written: stored by the render kernels next to the float image (camera-fed path), else one more pass
        img_u8 = (out["image_u8"] if "image_u8" in out else nerf.quantize_u8(out["image"])).reshape(ns, nv, hw, hw, 3)
        if multi:
            if pending["work"] is not None:
                wait_collective()
            pending["keep"] = img_u8                     # the source must stay alive until the collective has run
            pending["work"] = dist.all_gather_into_tensor(gathered[pending["i"] & 1], img_u8, async_op=True)
            pending["i"] += 1
        return out

    def wait_collective():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pending["work"].wait()                           # (the compute STREAM waits; the host does not)
        e1.record()
        pending["waits"].append((e0, e1))

    def drain():
        if pending["work"] is not None:
            wait_collective()
            pending["work"] = None
        nerf.finish_render(dec)                          # the last step's overflow flag (deferred check: nerf.render's docstring); inside the timed region

    def stat_pass(planes_, bits_):
        """one untimed pass for the integer statistics (exact sample count of this workload, hitting rays, boundary tests)"""
        render(planes_, bits_, want_counts=True)
        st = dec.last_render_stats
        counts = st["sample_counts"]
        res = dict(n_samples=int(counts.sum().item()), n_hit=int((counts > 0).sum().item()), overflow=int(st["overflow"].item()),
                   boundary=None if st.get("boundary_tests") is None else int(st["boundary_tests"].sum().item()))
        del counts
        return res

    def timed(planes_, bits_, warmup, steps, code_=None):
        events = []
        for _ in range(warmup):
            step(planes_, bits_, None, code_)
        drain()
        torch.cuda.synchronize()
        del pending["waits"][:]
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step(planes_, bits_, events, code_)
        drain()                                              # the last step's collective is inside the timed region
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if multi:
            # per-rank breakdown (r04): this rank's own wall time, its render launches and how long its compute stream stood waiting for collectives
            mine = torch.tensor([elapsed / steps * 1e3, float(np.mean([e[0].elapsed_time(e[2]) for e in events])),
                                 sum(a.elapsed_time(b) for a, b in pending["waits"]) / steps], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            pending["per_rank"] = [dict(rank=r, ms_per_step=float(v[0]), render_ms=float(v[1]), gather_wait_ms=float(v[2])) for r, v in enumerate(every)]
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, events, out

    stats = stat_pass(planes, bits)
    log(f"stat pass done: {stats['n_samples']} samples, overflow {stats['overflow']}, boundary tests {stats['boundary']}")
    code_dev = code_cpu.to(dev)                              # (only read if a batch must be redone through the stepwise path)
    # (r06) with next_batch, stage A of step i+1 runs on a second stream beside the shading kernel of step i: the events of the timed region then bracket the shading
    # kernel WITH that company (what `roofline` must report: the kernel's launches of the timed region) and no longer stage A.  A short untimed loop without the
    # prefetch, AHEAD of the timed region (whose launches stay the last ones of the process: tools/prof_render.sh), gives both stages alone, on this box, in this process
    alone = None
    prefetching = not (args.no_prefetch or args.packed_loop or rays is not None)
    if prefetching:
        args.no_prefetch = True
        _, ev_alone, _ = timed(planes, bits, 5, 10, code_dev)
        args.no_prefetch = False
        alone = dict(first_hit_ms=float(np.mean([e[0].elapsed_time(e[1]) for e in ev_alone])), shade_ms=float(np.mean([e[1].elapsed_time(e[2]) for e in ev_alone])))
        log(f"stages alone (no prefetch): first_hit {alone['first_hit_ms']:.3f} ms, shade {alone['shade_ms']:.3f} ms")
    elapsed, kernel_events, out = timed(planes, bits, args.warmup, args.steps, code_dev)
    per_rank = pending.get("per_rank")
    n_samples = stats["n_samples"]
    if multi:
        tot = torch.tensor([n_samples], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        n_samples_all = int(tot.item())
    else:
        n_samples_all = n_samples
    ms_per_step = elapsed / args.steps * 1e3
    log(f"timed region done: {ms_per_step:.2f} ms/step")
    rays_per_s = world * n_rays / (elapsed / args.steps)

    first_hit_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in kernel_events])) if alone is None else alone["first_hit_ms"]
    shade_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in kernel_events]))
    # dominant kernel = k_shade_mfma (gather + MLP + composite).  Its algorithmic bytes: 288 B per sample it shades plus,
    # per hitting ray, 8 B queue entry + 20 B outputs (+ 24 B ray when ray arrays are read).
    n_hit = stats["n_hit"]
    bytes_per_sample = BYTES_PER_SAMPLE if args.plane_dtype == "float32" else BYTES_PER_SAMPLE // 2     # 288 B per sample in fp32, 144 B with fp16 planes
    algo_bytes = n_samples * bytes_per_sample + n_hit * bytes_per_hit_ray
    achieved = algo_bytes / (shade_ms * 1e-3) / 1e9
    first_hit_bytes = n_rays * (24 if args.ray_arrays else 0) + (n_rays - n_hit) * 20 + n_hit * 8

    # HBM bytes per launch from the PMC counters: they cannot be read from inside the process that is being timed (rocprofv3 wraps the
    # whole command and a counter pass serialises every launch), so `traffic` is the figure of the most recent tools/prof_render.sh session
    # on THIS workload and kernel form; `traffic_source` says which session (file, the commit it profiled, the rocprofv3 launch average of the
    # same session) so that the number can be traced to profiles/
    traffic, traffic_source = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        w = tj["workload"]
        if (w["scenes"], w["views"], w["size"], w["variant"], w["plane_dtype"]) == (ns, nv, hw, args.variant, args.plane_dtype) and tj["kernel"].startswith("k_shade_mfma") \
                and w.get("ray_source", "arrays") == ("arrays" if args.ray_arrays else "cameras") \
                and int(tj.get("dir_products", 3)) == int(dec.shade_dir_products):
            traffic_source = {k: tj.get(k) for k in ("source", "profiled_commit", "kernel", "rocprof_launch_ms_avg_timed_steps", "rocprof_launches_averaged",
                                                     "hip_event_launch_ms_same_run", "method", "render_build_id")}
            from ssdnerf_amd.build import render_build_id
            mine = render_build_id()
            if tj.get("render_build_id") == mine:                    # (r06) the session profiled a library built from THESE render sources with THESE settings
                traffic = tj["hbm_bytes_per_launch"]
            else:
                traffic_source["refused"] = (f"the profiling session's library ({tj.get('render_build_id')}) is not the one timed here ({mine}): "
                                             "traffic is null until tools/prof_render.sh has been run on this build")
    except Exception:
        pass
    result = {
        "metric": "rays/s (rendered-views/s = rays/s / 16384), SRN Cars 128x128 novel-view render of cached triplanes",
        "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.plane_dtype == "float32" else "f32 math / f16 planes", "data": "synthetic",
        "config": {"workload": "ssdnerf_cars_uncond render of cached triplanes (BASELINE.json configs[1])", "scenes_per_gpu": ns,
                   "views_per_scene": nv, "image": f"{hw}x{hw}", "rays_per_step_per_gpu": n_rays, "grid_size": 64, "max_steps": 256,
                   "T_thresh": 1e-4, "dt_gamma": 0.0, "scene_variant": args.variant, "plane_dtype": args.plane_dtype, "parallelism": f"scene-parallel x{world}" + (" (TEST MODE: all ranks share cuda:0 over gloo)" if share_device else ""),
                   "ray_source": "(S,N,3) ray arrays" if args.ray_arrays else "cameras (rays generated in the kernels)",
                   "mlp_arithmetic": f"fp32 operands split into bf16 terms on the matrix cores: layer 1 all six products (2^-24 class), direction term {dec.shade_dir_products} of 6 "
                                     "(6 = the default since r04, fp32 class throughout; 3 = opt-in SSDNERF_SHADE_DIR_PRODUCTS=3, 2^-16 class on that additive term: "
                                     "see dir3_variant)",
                   "timed_call": "TriPlaneDecoder.render_packed(check_overflow=False) [--packed-loop]" if args.packed_loop else
                                 ("nerf.render (BaseNeRF.render on cached planes: two launches + the overflow-flag read per batch + uint8 views)" if args.sync_overflow_check else
                                  "nerf.render(defer_overflow_check=True) + nerf.finish_render after the last step (BaseNeRF.render on cached planes: two launches + uint8 "
                                  "views per batch; every batch's overflow flag is copied to pinned host memory asynchronously and examined behind the NEXT batch's "
                                  "launches -- a raised flag redoes that batch into its own output tensors; --sync-overflow-check restores the r04 read before the next launch)") +
                                 ("" if args.no_prefetch or args.packed_loop else "; next_batch: stage A of the next step is launched on a second stream beside this step's shading kernel (r06; --no-prefetch: the r05 step)"),
                   "collective": "all_gather(uint8 views), overlapped with the next step's render" if multi else "none"},
        "views_per_s": rays_per_s / (hw * hw), "samples_per_s": n_samples_all / (elapsed / args.steps),
        "mean_samples_per_ray": n_samples / n_rays, "rays_at_step_cap": stats["overflow"],
        "roofline": {"bound": "hbm", "kernel": "k_shade_mfma", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": algo_bytes,
                     "launch_ms": shade_ms, "launches_per_step": 1,
                     "launch_ms_alone": None if alone is None else alone["shade_ms"],
                     "frac_alone": None if alone is None else algo_bytes / (alone["shade_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "company": None if alone is None else "the timed region's launches share the chip with stage A of the NEXT step (second stream, nerf.render's next_batch): launch_ms / "
                                                           "frac are measured with that company; launch_ms_alone / frac_alone in a short loop without it (--no-prefetch times that loop)",
                     "note": f"algorithmic = {bytes_per_sample} B/sample + {bytes_per_hit_ray} B per hitting ray; planes (1.5 MiB/scene) are L2-resident, so real HBM "
                             "traffic is far below this (PMC numbers in DESIGN.md / profiles/)",
                     "other_kernels": {"first_hit (k_ray_cull + k_survivor_march)": {
                         "launch_ms": first_hit_ms, "hidden_beside_the_previous_steps_shading_kernel": prefetching, "algorithmic_bytes_per_launch": first_hit_bytes,
                         "achieved_GBs": first_hit_bytes / (first_hit_ms * 1e-3) / 1e9}}},
        "hit_rays_per_step_per_gpu": n_hit,
        "per_rank": None if per_rank is None else {
            "ranks": per_rank, "slowest_rank": max(per_rank, key=lambda r: r["ms_per_step"])["rank"],
            "note": "ms_per_step: the rank's own wall clock over the timed steps (the line's ms_per_step is the max); render_ms: HIP events around its two "
                    "render launches; gather_wait_ms: HIP events around every wait of its compute stream for an all-gather (0 when the collective hides "
                    "behind the next render)"},
        "boundary_rays": {"termination_tests_within_2e-6_of_T_thresh": stats["boundary"], "samples_per_step_per_gpu": n_samples},
    }

    extras = rank == 0 and not multi and not args.no_extras
    if extras:
        try:
            result["gpu_baseline"] = gpu_baseline_b1(dec, code_cpu[0].to(dev), bits[0], min(args.b1_views, nv), hw, out, nv, rays_per_s)
            log(f"gpu_baseline done: {result['gpu_baseline']['value']:.3g} rays/s")
        except Exception as e:                               # an extra must never cost the headline line
            result["gpu_baseline"] = {"error": repr(e)}
    if rank == 0 and not multi and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(params, code_cpu[0], bits[0].cpu().numpy(), min(args.cpu_views, nv), hw, out, nv)
        log("cpu_baseline done")
    del out
    if extras and args.variant == "object":
        try:
            code_u, planes_u, bits_u = build_scenes("uniform")
            st_u = stat_pass(planes_u, bits_u)
            el_u, ev_u, _ = timed(planes_u, bits_u, 1, 3, code_u.to(dev))
            result["uniform_variant"] = {"ms_per_step": el_u / 3 * 1e3, "rays_per_s": n_rays / (el_u / 3), "samples_per_s": st_u["n_samples"] / (el_u / 3),
                                         "mean_samples_per_ray": st_u["n_samples"] / n_rays,
                                         "shade_launch_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in ev_u])),
                                         "shade_algorithmic_GBs": (st_u["n_samples"] * BYTES_PER_SAMPLE + st_u["n_hit"] * bytes_per_hit_ray)
                                         / (float(np.mean([e[1].elapsed_time(e[2]) for e in ev_u])) * 1e-3) / 1e9,
                                         "boundary_tests": st_u["boundary"], "rays_at_step_cap": st_u["overflow"]}
            del planes_u, bits_u, code_u
            log("uniform variant done")
        except Exception as e:
            result["uniform_variant"] = {"error": repr(e)}
    if extras and args.variant == "object":
        # the opt-in precision class of the direction term beside the headline (r03 verdict weak #1: the default forms all six split products)
        try:
            other = 3 if dec.shade_dir_products == 6 else 6
            keep = dec.shade_dir_products
            dec.shade_dir_products = other
            try:
                el_d, ev_d, _ = timed(planes, bits, 3, 10, code_dev)
            finally:
                dec.shade_dir_products = keep
            ms_d = float(np.mean([e[1].elapsed_time(e[2]) for e in ev_d]))
            result[f"dir{other}_variant"] = {"direction_term_products": other, "ms_per_step": el_d / 10 * 1e3, "shade_launch_ms": ms_d,
                                              "roofline_frac": algo_bytes / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "note": "same workload, 10 timed steps; sample counts, depth and opacity are bit-identical between the two settings"}
            log(f"dir{other} variant done: shade {ms_d:.3f} ms")
        except Exception as e:
            result["dir_variant"] = {"error": repr(e)}
    del planes, code_dev
    torch.cuda.empty_cache()
    model = None
    if not args.no_extras:                                   # the sampling legs build the config's whole model (every rank at N > 1)
        try:
            model = build_model(dev, rank)
        except Exception as e:
            result["sampling"] = {"error": repr(e)}
        if multi:                                        # a rank that could not build the model must not leave the others in a collective
            ok = torch.tensor([0 if model is None else 1], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                model = None
                result.setdefault("sampling", {"error": "model build failed on another rank"})
    if extras and model is not None:
        try:
            result["ddim"] = ddim_leg(model, dev, ns, args.ddim_steps, log)
        except Exception as e:
            result["ddim"] = {"error": repr(e)}
    if model is not None:
        # north_star's node-level metric: scenes that complete DDIM + density grids + the 251-view render per second.
        # N == 1: fp32 (as the uncond configs run) and bf16 on rank 0.  N > 1: EVERY rank samples its own scenes (seeds 2021 + rank, as
        # --diff_seed does), renders them and takes part in the all-gather of the uint8 views; the time is the max over ranks.
        try:
            samp = {}
            for name in (("fp32", "bf16") if not multi else ("fp32",)):
                samp[name] = sampling_leg(model, dev, ns, nv, hw, args.ddim_steps, name, rank, world, dist if multi else None, log)
            if multi:
                # (r05) the N > 1 line carries the other two legs as well, on every rank: configs[3]'s 10-view form of the sampler, and -- below -- a
                # SHORTENED reconstruction (configs[2] / [4] shard their scenes over the ranks exactly like the sampler: lib/apis/test.py:12-73)
                samp["abo_10_views_fp32"] = sampling_leg(model, dev, ns, 10, hw, args.ddim_steps, "fp32", rank, world, dist, log)
            if not multi:
                # BASELINE.json configs[3] (ssdnerf_abotables_uncond.py:103-111): the same sampler, but val_uncond renders 10 views per scene, so the
                # leg is UNet-bound (the N > 1 form of this config is the `sampling` leg itself: every rank samples, renders and all-gathers)
                samp["abo_10_views_fp32"] = sampling_leg(model, dev, ns, 10, hw, args.ddim_steps, "fp32", rank, world, None, log)
                # a throughput-oriented batch: the configs sample 8 scenes per GPU per batch (samples_per_gpu=8), which leaves the UNet's low-resolution
                # half latency-bound (<= 32^2 pixels x 8 scenes per launch); 288 GB of HBM hold far more, and at 32 scenes per batch the same kernels
                # run 25-30 % faster per scene (tools/bench_unet.py --scenes 32: 12.1 ms bf16 / 28.2 ms fp32 per step).  Same sampler, same render.
                for name in ("fp32", "bf16"):
                    samp[f"batch32_{name}"] = sampling_leg(model, dev, 32, nv, hw, args.ddim_steps, name, rank, world, None, log)
            samp.update(scenes_per_s=samp["fp32"]["scenes_per_s"], config="uncond sampling as ssdnerf_cars_uncond runs it "
                        f"(fp32 UNet, {args.ddim_steps}-step DDIM, 8 density-grid refreshes, {nv} views of {hw}x{hw} per scene; abo_10_views_fp32: 10 views per "
                        "scene as ssdnerf_abotables_uncond renders), random UNet weights (fog-like scenes: the render leg's slow case)")
            # The UNet's weights are random, so the sampled scenes are fog and the render leg is its SLOW case (every ray marches through occupied
            # space: ~12 ms per scene against 0.77 ms for the object-like scenes of the headline step).  Derived, not measured: what each leg would
            # read with the headline step's render time per scene in place of the fog render -- the figure to expect from a trained prior.
            for leg in samp.values():
                if isinstance(leg, dict) and "ddim_ms" in leg and not multi:
                    t = (leg["ddim_ms"] + leg["density_ms"]) * 1e-3 + (ms_per_step * 1e-3 / ns) * leg["scenes_per_rank"] * leg["views_per_scene"] / nv
                    leg["derived_scenes_per_s_with_object_like_render"] = leg["scenes_per_rank"] / t
            result["sampling"] = samp
            result["scenes_per_s"] = samp["scenes_per_s"]
        except Exception as e:
            result["sampling"] = {"error": repr(e)}
    if extras and model is not None:
        try:
            result["recons"] = recons_leg(model, dev, ns, log, full=not args.no_full_recons)
        except Exception as e:
            result["recons"] = {"error": repr(e)}
    if multi and model is not None:
        # every rank reconstructs its own 8 scenes (no collective inside: scenes are independent); shortened -- 8 guided steps and 2 fine-tuning
        # iterations measured as differences, the 75 + 25 schedule projected -- so that the N = 1, 2, 4, 8 sweep stays within minutes
        try:
            rec = recons_leg(model, dev, ns, log, guide_steps=8, outer=2, full=False, seed=rank)
            mine = torch.tensor([rec["ms_per_guided_ddim_step"], rec["ms_per_finetune_iteration"], rec["projected_s_per_batch_75_guided_25_finetune"]],
                                dtype=torch.float64, device=dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            rec["per_rank"] = [dict(rank=r, ms_per_guided_ddim_step=float(v[0]), ms_per_finetune_iteration=float(v[1]), projected_s_per_batch=float(v[2]))
                               for r, v in enumerate(every)]
            slow = max(rec["per_rank"], key=lambda r: r["projected_s_per_batch"])
            rec.update(n_gpus=world, slowest_rank=slow["rank"], projected_scenes_per_s_all_ranks=world * ns / slow["projected_s_per_batch"],
                       note="shortened: 8 guided steps + 2 fine-tuning iterations per rank, schedule of configs[2] projected from the slowest rank")
            result["recons"] = rec
        except Exception as e:
            result["recons"] = {"error": repr(e)}
        if args.dry_nccl:
            result["dry_nccl"] = {"backend": dist.get_backend(), "world_size": world,
                                  "note": "N > 1 code path on ONE rank over RCCL: a self-check of initialisation, stream ordering and graph capture beside the "
                                          "collective library's threads -- not a scaling measurement"}
    if rank == 0:
        print(json.dumps(result))
    if multi:
        dist.destroy_process_group()


def gpu_baseline_b1(dec, code0, bits0, n_views, hw, gpu_out, nv, fused_rays_per_s):
    """the reference-shaped render on THIS GPU -- alive-ray loop, unfused march / composite operators, eager PyTorch decode,
    a host sync per iteration -- for `n_views` views of scene 0 as one ray batch (the reference batches all views of a scene)."""
    import numpy as np
    import torch
    from ssdnerf_amd import nerf, synthetic as S
    dev = code0.device
    poses = S.spiral_poses(nv)[:n_views].to(dev)[None]
    intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(1, n_views, -1)
    ro, rd = nerf.get_cam_rays(poses, intr, hw, hw)
    ro, rd = ro.reshape(1, -1, 3), rd.reshape(1, -1, 3)
    dec.render_mode, dec.eager_decode = "stepwise", True
    try:
        with torch.no_grad():
            for rep in range(2):                             # rep 0 warms the allocator / library kernel caches
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = dec(ro, rd, code0[None], bits0[None], 64, dt_gamma=0.0, perturb=False)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        iters = len(dec.last_render_stats["iterations"][0])
    finally:
        dec.render_mode, dec.eager_decode = "fused", False
    ws = res["weights_sum"][0]
    rgb = res["image"][0] + 1.0 * (1 - ws.unsqueeze(-1))
    got = gpu_out["image"][0].reshape(nv, hw * hw, 3)[:n_views].reshape(-1, 3)
    n = n_views * hw * hw
    return {"value": n / dt, "unit": "rays/s", "kind": "reference-shaped eager path on the same MI355X (B1)",
            "sample": f"scene 0, {n_views} views of {hw}x{hw} as one batch ({n} rays), {dt * 1e3:.0f} ms, {iters} loop iterations "
                      "(march_rays -> grid_sample + nn.Linear decode -> composite_rays -> compaction with a host sync each)",
            "speedup_of_fused_path": fused_rays_per_s / (n / dt), "max_abs_rgb_diff_vs_fused": float((rgb - got).abs().max().item())}


MODEL_CFG = dict(
    type="DiffusionNeRF", code_size=(3, 6, 128, 128), code_reshape=(18, 128, 128), code_activation=dict(type="TanhCode", scale=2), grid_size=64,
    diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"),
                   denoising=dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=128, channels_cfg=[1, 2, 2, 4, 4],
                                  resblocks_per_downsample=2, dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True,
                                  num_heads=4, attention_res=[32, 16, 8]),
                   timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5), denoising_mean_mode="V",
                   ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight", log_cfgs=None, data_info=dict(pred="v_t_pred", target="v_t"),
                                  weight_scale=4.0, scale_norm=True)),
    decoder=dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3],
                 use_dir_enc=True, dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256),
    decoder_use_ema=True, freeze_decoder=False, bg_color=1, pixel_loss=dict(type="MSELoss", loss_weight=20.0),
    reg_loss=dict(type="RegLoss", power=2, loss_weight=3e-3), cache_size=0, autocast_dtype=None,
    # test_cfg of configs/paper_cfgs/ssdnerf_cars_recons1v.py:79-97 (a superset of the uncond configs' keys; the legs set the step counts)
    test_cfg=dict(img_size=(128, 128), num_timesteps=50, clip_range=[-2, 2], density_thresh=0.1, dt_gamma_scale=0.5, n_inverse_rays=2 ** 14,
                  override_cfg={"diffusion_ema.ddpm_loss.weight_scale": 1.0}, loss_coef=0.1 / (128 * 128), guidance_gain=3.2 * (2 ** 14),
                  cond_mode="guide_optim", n_inverse_steps=25, extra_scene_step=3,
                  optimizer=dict(type="Adam", lr=0.005, weight_decay=0.0), lr_scheduler=dict(type="ExponentialLR", gamma=0.998)))


def build_model(dev, rank=0):
    """DiffusionNeRF of the cars configs (configs/paper_cfgs/ssdnerf_cars_uncond.py:3-61 / ssdnerf_cars_recons1v.py): 122 M-parameter UNet with
    random weights (no checkpoint is reachable: timing only), the synthetic decoder of the render bench."""
    import copy
    import torch
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODELS
    from ssdnerf_amd import synthetic as S
    model = MODELS.build(copy.deepcopy(MODEL_CFG))
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in model.diffusion_ema.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    model.decoder_ema.load_state_dict(S.make_decoder_params(2021), strict=False)
    return model.to(dev).eval()


def ddim_leg(model, dev, ns, n_steps, log):
    """50-step DDIM (eta 0, V-prediction, clip [-2, 2]) over `ns` scenes' (18,128,128) latents with the cars UNet (122 M parameters, random
    weights -- timing only), through the product's sampler: fp32 (what ssdnerf_cars_uncond runs: no autocast) and bf16 (config 5)."""
    import torch
    diff = model.diffusion_ema
    model.test_cfg["num_timesteps"] = diff.test_cfg["num_timesteps"] = n_steps
    g = torch.Generator(device=dev).manual_seed(0)
    out = {"scenes": ns, "steps": n_steps, "unet": "DenoisingUnetMod cars (122.4 M parameters, 218 GFLOP forward per scene)", "weights": "random"}
    noise = torch.randn(ns, 18, 128, 128, generator=g, device=dev)
    # parity at the benchmarked shape (r02 verdict weak #1a): one UNet evaluation on this batch by the eager fp32 module (library kernels) and by
    # each executor; tests/test_unet_fast_gpu.py::test_full_width_unet_matches_eager_at_the_bench_shape asserts the same quantity
    t_chk = torch.tensor([999, 979, 600, 339, 120, 59, 19, 0], device=dev)[:ns] if ns <= 8 else torch.full((ns,), 500, device=dev)
    with torch.no_grad():
        diff.denoising.fast_inference = False
        want = diff.denoising(noise, t_chk)
        diff.denoising.fast_inference = True
    for name, dt in (("fp32", None), ("bf16", torch.bfloat16)):
        best = None
        with torch.no_grad(), torch.autocast("cuda", enabled=dt is not None, dtype=dt):
            for rep in range(3):                             # rep 0: graph capture / library kernel selection
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                x0 = diff(noise, return_loss=False)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                if rep and (best is None or el < best):
                    best = el
        ms = best / n_steps * 1e3
        tflops = ns * UNET_FLOP_PER_SCENE / (ms * 1e-3) / 1e12
        ex = getattr(diff.denoising, "_fast_cache", {}).get(torch.float32 if dt is None else dt)
        with torch.no_grad(), torch.autocast("cuda", enabled=dt is not None, dtype=dt):
            got = diff.denoising(noise, t_chk).float()
        rel = float((got - want).norm() / want.norm())
        out[name] = {"ms_per_step": ms, "dtype": "f32 activations, bf16x2-split products on the matrix cores" if dt is None else "bf16",
                     "tflops": tflops, "mfma_frac_of_2.5PF": tflops / MFMA_PEAK_TFLOPS, "scenes_per_s": ns / best,
                     "library_fallback_ops": None if ex is None else getattr(ex, "library_fallbacks", None), "finite": bool(torch.isfinite(x0).all()),
                     "rel_err_vs_eager": rel}
        log(f"ddim {name}: {ms:.2f} ms/step, {tflops:.0f} TFLOP/s, rel err vs eager fp32 module {rel:.2e}")
    out.update(ms_per_step=out["fp32"]["ms_per_step"], dtype="fp32 config (ssdnerf_cars_uncond runs the UNet without autocast); bf16 beside it",
               tflops=out["fp32"]["tflops"], **{"mfma_frac_of_2.5PF": out["fp32"]["mfma_frac_of_2.5PF"]}, scenes_per_s=out["fp32"]["scenes_per_s"])
    return out


def sampling_leg(model, dev, ns, nv, hw, n_steps, dtype_name, rank, world, dist, log):
    """End-to-end unconditional sampling of `ns` scenes on this rank (lib/apis/test.py:12-73 evaluate_3d -> DiffusionNeRF.val_uncond + render,
    lib/models/autodecoders/diffusion_nerf.py:191-239): noise -> n_steps DDIM -> codes -> density grids (8 refreshes) -> nv views per scene ->
    uint8 quantisation (-> all-gather of the views at N > 1).  Scenes/s over ALL ranks; the time is the max over ranks."""
    import torch
    from ssdnerf_amd import nerf, synthetic as S
    model.autocast_dtype = None if dtype_name == "fp32" else "bfloat16"
    model.test_cfg["num_timesteps"] = model.diffusion_ema.test_cfg["num_timesteps"] = n_steps
    model.test_cfg["n_inverse_steps"] = 0                    # the uncond configs do not refine the samples
    poses = S.spiral_poses(nv).to(dev)[None].expand(ns, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(ns, nv, -1).contiguous()
    g = torch.Generator().manual_seed(2021 + rank)           # mirrors --diff_seed: distinct scenes per rank
    jit = [torch.rand(64 ** 3, 3, generator=g).to(dev) for _ in range(8)]
    gathered = torch.empty(world * ns, nv, hw, hw, 3, dtype=torch.uint8, device=dev) if dist is not None else None

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    best = None
    try:
        for rep in range(3):                                 # rep 0: graph capture, allocator warm-up
            noise = torch.randn(ns, 3, 6, 128, 128, generator=g).to(dev)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0 = ev()
            with torch.no_grad():
                with model._autocast():
                    latent = model.diffusion_ema(model.code_diff_pr(noise), return_loss=False)
                e1 = ev()
                code = model.code_diff_pr_inv(latent.float())
                _, bits = model.get_density(model.decoder_ema, code, cfg=model.test_cfg, jitters=jit)
                e2 = ev()
                image, _ = model.render(model.decoder_ema, code, bits, hw, hw, intr, poses, cfg=model.test_cfg)
                img_u8 = nerf.quantize_u8(image).reshape(ns, nv, hw, hw, 3)
                e3 = ev()
                if dist is not None:
                    dist.all_gather_into_tensor(gathered, img_u8)
            torch.cuda.synchronize()
            own = time.perf_counter() - t0                   # this rank's own time (the all-gather included), before it waits for the others
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            ranks = None
            if dist is not None:
                mine = torch.tensor([own * 1e3, e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)], dtype=torch.float64, device=dev)
                every = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(every, mine)
                ranks = [dict(rank=r, total_ms=float(v[0]), ddim_ms=float(v[1]), density_ms=float(v[2]), render_ms=float(v[3])) for r, v in enumerate(every)]
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            if rep and (best is None or el < best["total_s"]):
                best = dict(total_s=el, ddim_ms=e0.elapsed_time(e1), density_ms=e1.elapsed_time(e2), render_ms=e2.elapsed_time(e3),
                            finite=bool(torch.isfinite(image).all()))
                if ranks is not None:
                    best.update(per_rank=ranks, slowest_rank=max(ranks, key=lambda r: r["total_ms"])["rank"])
    finally:
        model.autocast_dtype = None
    out = dict(scenes_per_s=world * ns / best["total_s"], scenes_per_rank=ns, views_per_scene=nv, n_gpus=world, unet_dtype=dtype_name, **best)
    log(f"sampling {dtype_name}: {out['scenes_per_s']:.2f} scenes/s (ddim {best['ddim_ms']:.0f} ms, density {best['density_ms']:.0f} ms, render {best['render_ms']:.0f} ms)")
    return out


def recons_leg(model, dev, ns, log, guide_steps=3, outer=3, full=True, seed=0):
    """Config 3 (ssdnerf_cars_recons1v, cond_mode 'guide_optim'; lib/models/autodecoders/diffusion_nerf.py:241-311, 313-404): ms per rendering-guided
    DDIM step and ms per fine-tuning outer iteration (UNet forward + backward, then extra_scene_step + 1 train-branch render iterations) for `ns`
    scenes with one 128x128 conditioning view each, measured as (k + 1 iterations) - (1 iteration) so that fixed setup cancels; fp32 as the
    config runs it.  The 75 + 25 schedule of the config is projected from the two."""
    import numpy as np
    import torch
    from ssdnerf_amd import synthetic as S
    cfg = model.test_cfg
    saved = dict(cfg)
    g = torch.Generator().manual_seed(seed)
    codes = torch.stack([S.make_triplane(100 + seed * ns + i) for i in range(ns)]).to(dev)          # (distinct scenes per rank at N > 1)
    poses = S.spiral_poses()[[64]].to(dev)[None].expand(ns, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(128, 128).to(dev)[None, None].expand(ns, 1, -1).contiguous()
    with torch.no_grad():
        grid, bits = model.get_density(model.decoder_ema, codes, cfg=cfg)
        other = codes.roll(1, 0)                             # views of OTHER scenes: a loss with something to fit
        target, _ = model.render(model.decoder_ema, other, model.get_density(model.decoder_ema, other, cfg=cfg)[1], 128, 128, intr, poses, cfg=cfg)
    data = dict(cond_imgs=target.clamp(0, 1), cond_intrinsics=intr, cond_poses=poses)
    np.random.seed(0)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    def guide():
        with torch.enable_grad():
            return model.val_guide(dict(data, noise=torch.randn(ns, 3, 6, 128, 128, generator=g).to(dev)))

    def set_steps(n):
        cfg["num_timesteps"] = model.diffusion_ema.test_cfg["num_timesteps"] = n

    out = dict(scenes=ns, unet_dtype="fp32", cond_views_per_scene=1, rays_per_scene_per_iteration=2 ** 14)
    try:
        # warm-up: also long enough for the UNet's gradient path to capture its forward + backward graphs (unet.DenoisingUnetMod._grad_graph_call: after
        # `grad_graph_after` eager calls of a signature; a one-off like the executor's capture -- its cost is reported as `grad_graph.capture_s`)
        unet = model.diffusion_ema.denoising
        set_steps(2 + int(getattr(unet, "grad_graph_after", 0)))
        timed(guide)
        set_steps(1)
        _, t1 = timed(guide)
        set_steps(1 + guide_steps)
        _, tk = timed(guide)
        out["ms_per_guided_ddim_step"] = (tk - t1) / guide_steps * 1e3
        code_ = model.code_activation.inverse(codes)

        def optim():
            return model.val_optim(data, code_=code_.clone().requires_grad_(True), density_grid=grid.clone(), density_bitfield=bits.clone())

        cfg["n_inverse_steps"] = 1
        timed(optim)
        _, t1 = timed(optim)
        cfg["n_inverse_steps"] = 1 + outer
        (code, _, _), tk = timed(optim)
        out["ms_per_finetune_iteration"] = (tk - t1) / outer * 1e3
        out["inner_render_iterations_per_finetune_iteration"] = cfg["extra_scene_step"] + 1
        out["finite"] = bool(torch.isfinite(code).all())
        out["projected_s_per_batch_75_guided_25_finetune"] = (75 * out["ms_per_guided_ddim_step"] + 25 * out["ms_per_finetune_iteration"]) / 1e3
        out["projected_scenes_per_s"] = ns / out["projected_s_per_batch_75_guided_25_finetune"]
        out["grad_graph"] = unet.grad_graph_info() if hasattr(unet, "grad_graph_info") else None
        if full:
            out["full_batch"] = recons_full_batch(model, dev, ns, data, g, log, timed, prior_codes=codes)
            out["config5_full_batch"] = recons_full_batch(model, dev, ns, data, g, log, timed, config5=True, prior_codes=codes)
    finally:
        cfg.clear()
        cfg.update(saved)
        model.diffusion_ema.test_cfg["num_timesteps"] = saved["num_timesteps"]
    log(f"recons: {out['ms_per_guided_ddim_step']:.1f} ms per guided step, {out['ms_per_finetune_iteration']:.1f} ms per fine-tune iteration")
    return out


class _KnownScenePrior:
    """The synthetic PRIOR of the whole-batch reconstruction legs (r06; the r05 verdict's item 4).  A UNet with random weights is no prior: its V-prediction, or none at all
    (tools/recons_regime.py: output layers scaled by 1 .. 0), leaves the sampler with the initial noise as the code -- random triplanes decode to empty space, the guidance
    and the fine-tuning have nothing to hold on to, every view comes out white on every arithmetic and the r05 quality guard compared white with white.  This forward hook
    on the denoiser turns its output into  v = s * unet(x_t, t) + v*(x_t, t),  s = 0.05, where v* is the EXACT V-prediction of a Gaussian prior N(x0*, tau^2) around object
    codes x0* (one per scene of the batch; NOT the scenes of the conditioning views):  E[x0 | x_t] = x0* + k_t (x_t - a x0*),  k_t = a tau^2 / (a^2 tau^2 + b^2),
    v* = (a x_t - E[x0 | x_t]) / b.  At high noise it answers x0*, at low noise it follows x_t, so what the guidance writes into the latent survives to the next step.  The
    network still runs -- forward AND backward: the guidance gradient goes through it and through v* -- and contributes a perturbation of x0; the sampler lands near x0*,
    and the guided steps and the fine-tuning then have to move the codes from the prior's scene towards the one in the conditioning view.  Timing is unchanged (a few
    element-wise operations per call); the quality figures get something to measure."""

    def __init__(self, model, x0_star, scale=0.05, tau=0.5):
        self.diff, self.x0, self.scale, self.tau2 = model.diffusion_ema, model.code_diff_pr(x0_star).detach(), scale, tau * tau
        self.handle = self.diff.denoising.register_forward_hook(self)

    def __call__(self, module, inputs, output):
        import torch
        x_t, t = inputs[0], torch.as_tensor(inputs[1], device=inputs[0].device)
        if x_t.shape != self.x0.shape:
            return output
        if t.dim() == 0 or t.numel() != x_t.size(0):
            t = t.expand(x_t.size(0))
        ab = self.diff.schedule.signal_noise(x_t.device)[:, t]
        a, b = ab[0].reshape(-1, 1, 1, 1).to(output.dtype), ab[1].reshape(-1, 1, 1, 1).to(output.dtype)
        x = x_t.to(output.dtype)
        x0 = self.x0.to(output.dtype)
        x0_hat = x0 + (a * self.tau2 / (a * a * self.tau2 + b * b)) * (x - a * x0)
        return output * self.scale + (a * x - x0_hat) / b

    def remove(self):
        self.handle.remove()


def recons_full_batch(model, dev, ns, data, g, log, timed, config5=False, n_test_views=250, prior_codes=None):
    """One WHOLE reconstruction batch through ``DiffusionNeRF.val_step`` as the config ships it, wall-clock (the r03 verdict's missing #3: the
    75 + 25 figure was a projection from 3-step differences):
      * configs[2] ssdnerf_cars_recons1v (configs/paper_cfgs/ssdnerf_cars_recons1v.py:78-97,141): cond_mode 'guide_optim' = 75 rendering-guided DDIM
        steps, then 25 fine-tuning iterations of (prior loss through the UNet + 4 rendering-loss iterations), then the 250 test views of every scene;
      * config5 -- configs[4] ssdnerf_chairs_recons1v (ssdnerf_chairs_recons1v.py:79-97) in BASELINE.json's precision mix: the same plus 5 guided
        Langevin corrections (delta 0.4) after every DDIM step whose t_prev lies inside (0, 1000) -- 75 + 74 x 5 = 445 guided evaluations --,
        guidance_gain 0.4 * 2^14, snr_weight_power 0.25, ``autocast_dtype='bfloat16'`` and fp16 planes in the decoder (the 16-bit scene cache's layout).
    8 scenes, one 128x128 conditioning view each, synthetic targets, random UNet weights (timing + finiteness; parity of the pieces: tests/)."""
    import torch
    from ssdnerf_amd import synthetic as S
    cfg = model.test_cfg
    saved, saved_ac, saved_pd = dict(cfg), model.autocast_dtype, model.decoder_ema.plane_dtype
    saved_d = dict(model.diffusion_ema.test_cfg)               # (the diffusion module holds its own copy of test_cfg)
    poses = S.spiral_poses(251)[:n_test_views].to(dev)[None].expand(ns, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(128, 128).to(dev)[None, None].expand(ns, n_test_views, -1).contiguous()
    try:
        cfg.update(num_timesteps=75, n_inverse_steps=25, extra_scene_step=3, cond_mode="guide_optim")
        model.diffusion_ema.test_cfg.update(num_timesteps=75)
        if config5:
            extra = dict(langevin_steps=5, langevin_delta=0.4, guidance_gain=0.4 * (2 ** 14), snr_weight_power=0.25)
            cfg.update(extra)
            model.diffusion_ema.test_cfg.update(extra)
            model.autocast_dtype = "bfloat16"
            model.decoder_ema.plane_dtype = torch.float16
        n_eval = len(model.diffusion_ema.sampling_plan("ddim"))

        import numpy as np
        noise = torch.randn(ns, 3, 6, 128, 128, generator=g).to(dev)

        def run():
            # every draw of the batch is seeded the same way for the timed run and for the reference run below: the initial noise is shared, the
            # host-side draws (prior-loss noise and Langevin noise: torch's CPU generator; timesteps: numpy) and the device draws (march jitter,
            # density-grid jitter: torch's device generator) restart from fixed seeds
            torch.manual_seed(1234); np.random.seed(1234)
            return model.val_step(dict(data, noise=noise, test_poses=poses, test_intrinsics=intr))

        def psnr(a, b):                                          # lib/core/evaluation/metrics.py:52-55 (eval_psnr), per image, then the mean
            mse = (a.float() - b.float()).square().flatten(-3).mean(dim=-1)
            return 10.0 * (-torch.log10(mse + 1e-6))

        prior = _KnownScenePrior(model, 0.8 * prior_codes) if prior_codes is not None else None
        res, wall = timed(run)
        ok = bool(torch.isfinite(res["code"]).all()) and bool(torch.isfinite(res["pred_imgs"]).all())
        # what the guided steps alone reach (cond_mode 'guide': no fine-tuning), untimed: the conditioning view's PSNR must RISE from there to the full batch's
        guided_only = None
        if prior is not None and n_test_views > 64:
            cfg["cond_mode"] = "guide"
            try:
                r0 = run()
                guided_only = r0["pred_imgs"][:, 64].clone()
                del r0
            finally:
                cfg["cond_mode"] = "guide_optim"
        pred = res["pred_imgs"]                                  # (ns, views, 3, h, w), quantised to k / 255 like the reference's eval_and_viz
        # (a) against the conditioning view: test view 64 IS the conditioning pose, its image should reproduce the target the guidance was given
        cond = data["cond_imgs"][:, 0].permute(0, 3, 1, 2) if data["cond_imgs"].dim() == 5 else None
        psnr_cond = psnr(pred[:, 64], cond) if cond is not None and n_test_views > 64 else None
        psnr_cond_guided = psnr(guided_only, cond) if cond is not None and guided_only is not None else None
        psnr_cond_prior = None
        if prior is not None and cond is not None and n_test_views > 64:      # ... and the prior's own scene seen from the conditioning pose: where the batch starts from
            with torch.no_grad():
                pc = (0.8 * prior_codes).to(dev)
                img0, _ = model.render(model.decoder_ema, pc, model.get_density(model.decoder_ema, pc, cfg=cfg)[1], 128, 128, data["cond_intrinsics"], data["cond_poses"], cfg=cfg)
                psnr_cond_prior = psnr((torch.round(img0[:, 0].clamp(0, 1) * 255) / 255).permute(0, 3, 1, 2), cond)
        # (b) the same batch, same seeds, through the REFERENCE-SHAPED arithmetic once, untimed: fp32, no autocast, fp32 planes, the eager modules
        # (no captured graphs, no inference executor) -- what the fast path has to agree with.  `parity` is false below 35 dB.
        # r05 (last): the reference run also leaves this repository's UNet / decode-gradient KERNELS -- its convolutions, GroupNorm and attention go through
        # PyTorch's library operators in IEEE fp32 (MIOpen / native kernels, forward and backward by autograd), the triplane decode's gradient through autograd
        # over the reference-shaped decode -- so the two runs share the ray march / compositing operators (pinned bit for bit elsewhere) and nothing else.
        unet = model.diffusion_ema.denoising
        from ssdnerf_amd import unet as U
        dec_cls = type(model.decoder_ema)
        saved_fast = (getattr(unet, "fast_inference", None), getattr(unet, "grad_graph", None), model.autocast_dtype, model.decoder_ema.plane_dtype)
        saved_lib = (U._Conv2d.grad_conv, U.GRAD_ATT_KERNEL, U.GRAD_ATT_POINTWISE, U.GRAD_GN, U.GRAD_ATT, dec_cls.fused_code_grad)
        lib_before = U._Conv2d.library_calls
        try:
            if saved_fast[0] is not None:
                unet.fast_inference = False
            if saved_fast[1] is not None:
                unet.grad_graph = False
            U._Conv2d.grad_conv = False
            U.GRAD_ATT_KERNEL = U.GRAD_ATT_POINTWISE = U.GRAD_GN = U.GRAD_ATT = False
            dec_cls.fused_code_grad = False
            model.autocast_dtype = None
            model.decoder_ema.plane_dtype = torch.float32
            ref_error = None
            try:
                ref, ref_wall = timed(run)
            except Exception as e:                                   # (the untimed checker must not take the bench line down: fall back to the eager modules on this
                ref_error = f"{type(e).__name__}: {e}"[:300]         # repository's gradient-path kernels, the r05 form of this comparison, and say so)
                log(f"library-arithmetic reference run failed ({ref_error}); repeating it on the eager modules with the gradient-path kernels")
                U._Conv2d.grad_conv, U.GRAD_ATT_KERNEL, U.GRAD_ATT_POINTWISE, U.GRAD_GN, U.GRAD_ATT, dec_cls.fused_code_grad = saved_lib
                ref, ref_wall = timed(run)
        finally:
            if prior is not None:
                prior.remove()
            if saved_fast[0] is not None:
                unet.fast_inference = saved_fast[0]
            if saved_fast[1] is not None:
                unet.grad_graph = saved_fast[1]
            U._Conv2d.grad_conv, U.GRAD_ATT_KERNEL, U.GRAD_ATT_POINTWISE, U.GRAD_GN, U.GRAD_ATT, dec_cls.fused_code_grad = saved_lib
            model.autocast_dtype, model.decoder_ema.plane_dtype = saved_fast[2], saved_fast[3]
        ref_library_convs = U._Conv2d.library_calls - lib_before      # > 0: the reference's gradient calls really went to the library convolution
        psnr_ref = psnr(pred, ref["pred_imgs"])                  # (ns, views)
        # how much of the views is not background (bg 1.0): a reconstruction that came out empty would make both PSNR figures meaningless
        foreground = float((pred < 0.995).any(dim=2).float().mean())
        identical = float((pred == ref["pred_imgs"]).all(dim=2).float().mean())
        code_err = float((res["code"].float() - ref["code"].float()).abs().max())
        # the same comparison on the CODES (what the guided steps and the fine-tuning produce; the views are a function of them): PSNR with the reference
        # code's own range as peak.  With random UNet weights the fine-tuning of this synthetic batch ends in EMPTY scenes (see foreground_pixel_fraction:
        # white views on both sides, image PSNR at its cap), so the code figure is the one that can move
        code_scale = float(ref["code"].float().abs().max())
        code_mse = float((res["code"].float() - ref["code"].float()).square().mean())
        code_psnr = 10.0 * math.log10(code_scale ** 2 / max(code_mse, 1e-20))
        parity = bool(psnr_ref.mean() >= 35.0) and code_psnr >= 35.0
        out = dict(wall_s=wall, scenes_per_s=ns / wall, guided_unet_evaluations=n_eval, finetune_outer_iterations=25, inner_render_iterations=4,
                   test_views_per_scene=n_test_views, finite=ok,
                   psnr_conditioning_view_db=None if psnr_cond is None else dict(mean=float(psnr_cond.mean()), min=float(psnr_cond.min())),
                   psnr_conditioning_view_after_guidance_only_db=None if psnr_cond_guided is None else dict(mean=float(psnr_cond_guided.mean()), min=float(psnr_cond_guided.min())),
                   psnr_conditioning_view_of_the_priors_scene_db=None if psnr_cond_prior is None else dict(mean=float(psnr_cond_prior.mean()), min=float(psnr_cond_prior.min())),
                   synthetic_prior=None if prior is None else "v = 0.05 * UNet(x_t, t) + v*(x_t, t): the random-weight UNet (forward and backward) plus the exact V-prediction of a Gaussian "
                                                              "prior N(x0*, 0.5^2) around object scenes x0* -- other scenes than the conditioning views show (bench.py: _KnownScenePrior)",
                   psnr_vs_fp32_eager_reference_db=dict(mean=float(psnr_ref.mean()), min=float(psnr_ref.min()), views=int(psnr_ref.numel())),
                   max_abs_code_diff_vs_reference=code_err, code_abs_max=code_scale, code_psnr_vs_reference_db=code_psnr, reference_wall_s=ref_wall, parity=parity,
                   foreground_pixel_fraction=foreground, pixels_identical_to_reference_fraction=identical,
                   code_fraction_above_1p9=float((res["code"].float().abs() > 1.9).float().mean()),      # (clip_range is +-2: how much of the code sits in the clamp)
                   precision="bf16 autocast (UNet) + fp16 planes" if config5 else "fp32",
                   reference_arithmetic="PyTorch library operators in fp32 for the UNet (MIOpen convolutions, native GroupNorm / attention, autograd backward), autograd over the "
                                        "reference-shaped triplane decode, fp32 planes, no autocast, no executor, no captured graphs; shared with the timed run: the ray "
                                        "march / compositing operators and the fused render of the test views",
                   reference_library_convolution_calls=ref_library_convs, reference_error=ref_error,
                   note="one timed val_step (guide_optim) after the warm-up the per-step measurements above provide; then the SAME batch (same noise, same "
                        "seeds) once through the library-arithmetic reference (reference_arithmetic), untimed: PSNR of the 250 views per scene between the two "
                        "(lib/core/evaluation/metrics.py:52-55), and of the codes.  psnr_conditioning_view_*: test view 64 (the conditioning pose) against the target -- of the "
                        "prior's own scene (where the batch starts), after the 75 guided steps alone (cond_mode 'guide', untimed), after the whole batch.  synthetic_prior: "
                        "r05 ran the random-weight UNet bare and every arithmetic reconstructed EMPTY scenes (white against white); r06 adds the analytic term of a prior "
                        "that knows object scenes, so that the codes stay object-like and un-saturated (code_abs_max, foreground_pixel_fraction) and the figures can move; "
                        "`parity` is false below 35 dB on views or codes")
    finally:
        cfg.clear(); cfg.update(saved)
        model.diffusion_ema.test_cfg.clear(); model.diffusion_ema.test_cfg.update(saved_d)
        model.autocast_dtype, model.decoder_ema.plane_dtype = saved_ac, saved_pd
    log(f"recons {'config 5' if config5 else 'config 3'} full batch: {wall:.2f} s for {ns} scenes ({n_eval} guided evaluations + 25 x 5 fine-tune); "
        f"PSNR vs fp32 eager {out['psnr_vs_fp32_eager_reference_db']['mean']:.1f} dB (min {out['psnr_vs_fp32_eager_reference_db']['min']:.1f}), "
        f"conditioning view {out['psnr_conditioning_view_db']}")
    return out


def cpu_baseline(params, code0, bits0, n_views, hw, gpu_out, nv):
    """Oracle on the host cores: scene 0, `n_views` views, same triplane / bitfield / rays as the GPU run.
    Also reports the GPU-vs-oracle error on that sample."""
    import numpy as np
    import torch
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
