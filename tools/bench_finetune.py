#!/usr/bin/env python
"""tools/bench_finetune.py -- single-view reconstruction on one MI355X (BASELINE.json configs[2], ssdnerf_cars_recons1v): rendering-guided
DDIM steps (``val_guide``) followed by fine-tuning with the diffusion prior (``val_optim``).
Prints one JSON line with ms per guided DDIM step and ms per fine-tuning outer iteration (= 1 UNet forward+backward + (extra_scene_step+1)
train-branch render forward+backward + optimizer steps) for ``--scenes`` scenes.  Random weights, synthetic decoder and target: timing only."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ssdnerf_amd  # noqa
from ssdnerf_amd.registry import MODELS
from ssdnerf_amd import synthetic as S

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--guide-steps", type=int, default=3); ap.add_argument("--outer", type=int, default=3)
ap.add_argument("--extra-scene-step", type=int, default=3); ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
ap.add_argument("--full-batch", action="store_true", help="also time (and with --cprofile: profile the host side of) ONE whole val_step: 75 guided steps + 25 x (1 + 4) + 250 views")
ap.add_argument("--cprofile", action="store_true", help="cProfile of the timed fine-tuning call (host side: top functions by own time)")
ap.add_argument("--aten", action="store_true", help="with --profile: the aten:: operators with device time by input shape (copies, fills, accumulations outside the custom kernels)")
ap.add_argument("--profile", action="store_true", help="instead of timing: torch.profiler tables (top kernels by device time) of one guided step and one outer iteration")
a = ap.parse_args()
cfg = dict(type="DiffusionNeRF", code_size=(3, 6, 128, 128), code_reshape=(18, 128, 128), code_activation=dict(type="TanhCode", scale=2), grid_size=64,
           diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"),
                          denoising=dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=128, channels_cfg=[1, 2, 2, 4, 4],
                                         resblocks_per_downsample=2, dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True,
                                         num_heads=4, attention_res=[32, 16, 8]),
                          timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5), denoising_mean_mode="V",
                          ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight",
                                         log_cfgs=dict(type="quartile", prefix_name="loss_mse", total_timesteps=1000),
                                         data_info=dict(pred="v_t_pred", target="v_t"), weight_scale=4.0, scale_norm=True)),
           decoder=dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3],
                        use_dir_enc=True, dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256),
           decoder_use_ema=True, freeze_decoder=False, bg_color=1, pixel_loss=dict(type="MSELoss", loss_weight=20.0),
           reg_loss=dict(type="RegLoss", power=2, loss_weight=3e-3), cache_size=0, autocast_dtype=dict(fp32=None, bf16="bfloat16")[a.dtype],
           test_cfg=dict(img_size=(128, 128), num_timesteps=75, clip_range=[-2, 2], density_thresh=0.1, dt_gamma_scale=0.5, n_inverse_rays=2 ** 14,
                         override_cfg={"diffusion_ema.ddpm_loss.weight_scale": 1.0}, loss_coef=0.1 / (128 * 128), guidance_gain=3.2 * (2 ** 14),
                         cond_mode="guide_optim", n_inverse_steps=a.outer, extra_scene_step=a.extra_scene_step,
                         optimizer=dict(type="Adam", lr=0.005, weight_decay=0.0), lr_scheduler=dict(type="ExponentialLR", gamma=0.998)))
model = MODELS.build(cfg)
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for p in model.diffusion_ema.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * 0.02)
model.decoder_ema.load_state_dict(S.make_decoder_params(), strict=False)
model = model.cuda().eval()
ns = a.scenes
codes = torch.stack([S.make_triplane(100 + i) for i in range(ns)]).cuda()
poses = S.spiral_poses()[[64]].cuda()[None].expand(ns, -1, -1, -1).contiguous()
intr = S.cars_intrinsics(128, 128).cuda()[None, None].expand(ns, 1, -1).contiguous()
with torch.no_grad():
    grid, bits = model.get_density(model.decoder_ema, codes, cfg=model.test_cfg)
    target, _ = model.render(model.decoder_ema, codes.roll(1, 0), model.get_density(model.decoder_ema, codes.roll(1, 0), cfg=model.test_cfg)[1],
                             128, 128, intr, poses, cfg=model.test_cfg)       # views of OTHER scenes: a loss with something to fit
data = dict(cond_imgs=target.clamp(0, 1), cond_intrinsics=intr, cond_poses=poses)
np.random.seed(0)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, time.perf_counter() - t0


if a.profile:
    from torch.profiler import profile, ProfilerActivity
    code_ = model.code_activation.inverse(codes)
    model.test_cfg["num_timesteps"] = 1; model.diffusion_ema.test_cfg["num_timesteps"] = 1; model.test_cfg["n_inverse_steps"] = 1
    runs = dict(guided_ddim_step=lambda: model.val_guide(dict(data, noise=torch.randn(ns, 3, 6, 128, 128, generator=g).cuda())),
                finetune_outer_iteration=lambda: model.val_optim(data, code_=code_.clone().requires_grad_(True), density_grid=grid.clone(),
                                                                 density_bitfield=bits.clone()))
    for name, fn in runs.items():
        timed(fn)                                                             # warm-up
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=a.aten) as prof:
            _, dt = timed(fn)
        print(f"==== {name}: {dt * 1e3:.1f} ms wall (includes one-off setup: rays, optimizer, grids)")
        if a.aten:
            rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
            print(f"aten operators with device time: {sum(e.self_device_time_total for e in rows) * 1e-3:.3f} ms")
            for e in sorted(rows, key=lambda e: -e.self_device_time_total)[:45]:
                print(f"  {e.key:30s} x{e.count:3d} {e.self_device_time_total * 1e-3:7.3f} ms  {str(e.input_shapes)[:140]}")
            continue
        print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=70))
    sys.exit(0)

out = dict(scenes=ns, unet_dtype=a.dtype)
# guidance: time k and 1 step runs, the difference is the per-step cost without the fixed setup
unet = model.diffusion_ema.denoising
model.test_cfg["num_timesteps"] = model.diffusion_ema.test_cfg["num_timesteps"] = 2 + int(getattr(unet, "grad_graph_after", 0))
timed(lambda: model.val_guide(dict(data, noise=torch.randn(ns, 3, 6, 128, 128, generator=g).cuda())))      # warm-up (long enough for the gradient path's graph capture)
model.test_cfg["num_timesteps"] = 1
model.diffusion_ema.test_cfg["num_timesteps"] = 1
_, t1 = timed(lambda: model.val_guide(dict(data, noise=torch.randn(ns, 3, 6, 128, 128, generator=g).cuda())))
model.test_cfg["num_timesteps"] = 1 + a.guide_steps
model.diffusion_ema.test_cfg["num_timesteps"] = 1 + a.guide_steps
if a.cprofile:
    import cProfile, pstats
    prg = cProfile.Profile(); prg.enable()
_, tk = timed(lambda: model.val_guide(dict(data, noise=torch.randn(ns, 3, 6, 128, 128, generator=g).cuda())))
if a.cprofile:
    prg.disable(); print("==== guided steps (host side)"); pstats.Stats(prg).sort_stats("tottime").print_stats(22)
out["ms_per_guided_ddim_step"] = round((tk - t1) / a.guide_steps * 1e3, 2)

code_ = model.code_activation.inverse(codes)
model.test_cfg["n_inverse_steps"] = 1
timed(lambda: model.val_optim(data, code_=code_.clone().requires_grad_(True), density_grid=grid.clone(), density_bitfield=bits.clone()))   # warm-up
_, t1 = timed(lambda: model.val_optim(data, code_=code_.clone().requires_grad_(True), density_grid=grid.clone(), density_bitfield=bits.clone()))
model.test_cfg["n_inverse_steps"] = 1 + a.outer
if a.cprofile:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
(code, _, _), tk = timed(lambda: model.val_optim(data, code_=code_.clone().requires_grad_(True), density_grid=grid.clone(), density_bitfield=bits.clone()))
if a.cprofile:
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(28)
out["ms_per_finetune_outer_iteration"] = round((tk - t1) / a.outer * 1e3, 2)
out["inner_render_iterations_per_outer"] = a.extra_scene_step + 1
out["code_finite"] = bool(torch.isfinite(code).all())
# the recons1v schedule: 75 guided steps + 25 outer iterations
out["projected_s_per_batch_75_guided_25_outer"] = round((75 * out["ms_per_guided_ddim_step"] + 25 * out["ms_per_finetune_outer_iteration"]) / 1e3, 2)
out["grad_graph"] = unet.grad_graph_info() if hasattr(unet, "grad_graph_info") else None
if a.full_batch:
    tp = S.spiral_poses(251)[:250].cuda()[None].expand(ns, -1, -1, -1).contiguous()
    ti = S.cars_intrinsics(128, 128).cuda()[None, None].expand(ns, 250, -1).contiguous()
    model.test_cfg.update(num_timesteps=75, n_inverse_steps=25, extra_scene_step=3, cond_mode="guide_optim")
    model.diffusion_ema.test_cfg.update(num_timesteps=75)
    run = lambda: model.val_step(dict(data, noise=torch.randn(ns, 3, 6, 128, 128, generator=g).cuda(), test_poses=tp, test_intrinsics=ti))
    if a.cprofile:
        import cProfile, pstats
        prf = cProfile.Profile(); prf.enable()
    _, tb = timed(run)
    if a.cprofile:
        prf.disable(); print("==== whole batch (host side)"); st_ = pstats.Stats(prf).sort_stats("tottime"); st_.print_stats(30); st_.print_callers("item"); st_.print_callers("method .to. of"); st_.print_callers("run_backward")
    out["full_batch_s"] = round(tb, 3)
print(json.dumps(out))
